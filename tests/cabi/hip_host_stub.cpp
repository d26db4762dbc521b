// hip_host_stub.cpp — TEST INFRASTRUCTURE, not product code.
//
// A stand-in for libamdhip64 for the CPU container: the HOST halves of csrc/*.hip (compiled with `hipcc --cuda-host-only -fsanitize=…`) link against it so
// that the library's host-side bookkeeping — contexts, segment offsets and sizes, buffer lifetimes, the process-wide state several rank threads share —
// runs under AddressSanitizer and ThreadSanitizer (SURVEY.md §5 "sanitizers"; GPU sanitizers are not available on the pool).  "Device" memory is host
// memory (calloc: zero-filled, so counts read back as 0), copies are memcpy — ASan therefore checks every copy's bounds against the allocation it
// touches —, kernel launches do NOTHING: no result is meaningful and nothing here says anything about parity.  One device, named gfx950.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "hip_host_stub"; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t* p, int) { memset(p, 0, sizeof *p); strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-"); p->multiProcessorCount = 256; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) { *v = a == hipDeviceAttributeMultiprocessorCount ? 256 : 0; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemcpyToSymbol(const void*, const void*, size_t, size_t, hipMemcpyKind) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipFuncGetAttributes(hipFuncAttributes* a, const void*) { memset(a, 0, sizeof *a); return hipSuccess; }
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* s) { *s = hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p) { memset(h, 0, sizeof *h); memcpy(h, &p, sizeof p); return hipSuccess; }
hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned) { memcpy(p, &h, sizeof *p); return hipSuccess; }   // (ranks are threads of one process here)
hipError_t hipIpcCloseMemHandle(void*) { return hipSuccess; }
// kernel launches: the host-only objects call these; nothing runs
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { return hipSuccess; }
hipError_t hipExtLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t, hipEvent_t, hipEvent_t, int) { return hipSuccess; }
static thread_local struct { dim3 g, b; size_t shm; hipStream_t st; } g_cfg;
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t shm, hipStream_t st) { g_cfg.g = g; g_cfg.b = b; g_cfg.shm = shm; g_cfg.st = st; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* shm, hipStream_t* st) { *g = g_cfg.g; *b = g_cfg.b; *shm = g_cfg.shm; *st = g_cfg.st; return hipSuccess; }
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
}
