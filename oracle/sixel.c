/* placeholder, filled in below */
#include "timg_oracle.h"
