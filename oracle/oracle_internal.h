/* oracle/oracle_internal.h -- TEST INFRASTRUCTURE ONLY. Shared helpers. */
#ifndef TIMG_ORACLE_INTERNAL_H
#define TIMG_ORACLE_INTERNAL_H
#include <stdint.h>

typedef struct {
    float r, g, b, a;
} lin_t; /* timg::LinearColor, src/framebuffer.h:138-174 */

lin_t lin_from_rgba(const uint8_t *p);
void lin_repack(const lin_t *l, uint8_t *out);
uint8_t term256(const uint8_t *p);

#endif
