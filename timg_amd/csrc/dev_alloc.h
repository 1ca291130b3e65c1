// timg_amd/csrc/dev_alloc.h -- every device allocation of libtimg_hip.so goes through here.
//
// Normally DevMalloc is hipMalloc.  With TIMG_HIP_GUARD=start|end16|end4 in the environment (a TEST
// aid, read once) an allocation gets its own virtual range with an UNMAPPED granule on either side
// (hipMemAddressReserve / hipMemMap), is placed flush against one of them and the mapped slack is
// poisoned:
//   start : the buffer begins at the first mapped byte          (reads/writes BEFORE it fault)
//   end16 : the buffer ends at most 15 bytes before the unmapped granule, 16-byte aligned start
//   end4  : the buffer ends exactly at the unmapped granule (sizes are rounded up to 4 bytes)
// A kernel that reads or writes past a buffer therefore dies with a GPU memory access fault instead
// of silently touching a neighbour, and DevFree aborts when the poison around the buffer changed.
// tests/test_zy_guard_pages.py runs the GPU parity suites in child processes in each mode.
#ifndef TIMG_AMD_DEV_ALLOC_H
#define TIMG_AMD_DEV_ALLOC_H

#include <hip/hip_runtime.h>

#include <cstddef>

namespace timg_amd {
hipError_t DevMalloc(void **ptr, size_t bytes);
hipError_t DevFree(void *ptr);
// 0 = off, 1 = start, 2 = end16, 3 = end4
int GuardMode();
// TIMG_HIP_FAIL_MALLOC=<k> (a TEST aid): the k-th DevMalloc after the first call of this function fails once with
// hipErrorOutOfMemory.  timg_hip_init calls it when a context exists.
void ArmMallocInjection();

}  // namespace timg_amd

#endif
