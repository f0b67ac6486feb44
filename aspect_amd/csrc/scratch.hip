// scratch.hip -- the library's static device scratch, for callers that may not allocate
// (agmg_vmult_device_noalloc, include/agmg.h).  One file-scope __device__ array per loaded code
// object, i.e. one copy per GPU, held until the process ends.  Host code never passes the array's
// name to a kernel: the name is a host-side stand-in; the device address of each GPU's copy comes
// from hipGetSymbolAddress, once per device, with that device current (agmg_create).
#include "agmg_internal.h"

namespace agmg {

namespace {
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];
constexpr int MAX_DEVICES = 64;
void *g_scratch_ptr[MAX_DEVICES] = {nullptr};
}  // namespace

// call with `device` current; idempotent
int static_scratch_init(int device) {
  if (device < 0 || device >= MAX_DEVICES) { set_error("static scratch: device ordinal out of range"); return -1; }
  if (g_scratch_ptr[device]) return 0;
  void *p = nullptr;
  AGMG_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
  if (!p) { set_error("static scratch: hipGetSymbolAddress returned a null address"); return -100; }
  g_scratch_ptr[device] = p;
  return 0;
}

// device address of `device`'s copy of the array (nullptr before static_scratch_init)
void *static_scratch_ptr(int device) {
  return (device >= 0 && device < MAX_DEVICES) ? g_scratch_ptr[device] : nullptr;
}

}  // namespace agmg
