// Test-only library: evaluates the exact-arithmetic device helpers of the HIP path
// (raytracing_amd/csrc/lt_math.h) on arrays, so that known-answer tests can compare them
// with integer arithmetic done on the host.  Not part of libltgpu.so.
#include "../../raytracing_amd/csrc/lt_math.h"
#include <stdint.h>

using namespace lt;

enum { OP_TDIVR = 0, OP_UDIV_R = 1, OP_PERSP_DIV = 2, OP_TDIV_ANY = 3, OP_TI_SQRT = 4, OP_UNIT_COMPONENT = 5, OP_TDIV = 6 };

__global__ void __launch_bounds__(256) kat_kernel(int op, const long long *__restrict__ a, const long long *__restrict__ b,
                                                  const long long *__restrict__ c, long long *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    long long r = 0;
    switch (op) {
    case OP_TDIVR: r = tdivr((int)a[i], drcp((int)b[i])); break;
    case OP_UDIV_R: r = udiv_r((int)a[i], (int)b[i], __builtin_amdgcn_rcpf((float)(int)b[i])); break;
    case OP_PERSP_DIV: r = persp_div((int)a[i], (int)c[i], (int)b[i], drcp((int)b[i])); break;
    case OP_TDIV_ANY: r = tdiv_any((int)a[i], (int)b[i]); break;
    case OP_TI_SQRT: r = (long long)ti_sqrt((unsigned long long)a[i]); break;
    case OP_UNIT_COMPONENT: r = unit_component((int)a[i], b[i]); break;
    case OP_TDIV: r = tdiv((int)a[i], (int)b[i]); break;
    }
    out[i] = r;
}

// a, b, c, out: device arrays of n int64 (unused operands may be null); returns the hipError_t
extern "C" int lt_kat_eval(int op, const long long *a, const long long *b, const long long *c, long long *out, size_t n)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(kat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, a, b, c, out, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    return (int)hipDeviceSynchronize();
}

// Fills the LDS of every CU with `value` (as many 64 KB workgroups as fit the chip, several
// times over): what a later kernel finds in LDS words it has not written itself.
__global__ void __launch_bounds__(256) lds_fill_kernel(int value, int *sink)
{
    extern __shared__ int lds_words[];
    const int n = 64 * 1024 / 4;
    for (int i = threadIdx.x; i < n; i += 256) lds_words[i] = value;
    __syncthreads();
    // keep the stores alive, and the workgroup resident for a moment so that others land on other CUs
    int s = 0;
    for (int i = threadIdx.x; i < n; i += 256 * 64) s += lds_words[i];
    if (s == 0x7fffffff) *sink = s;
}

extern "C" int lt_kat_fill_lds(int value, int *d_sink)
{
    hipError_t e = hipFuncSetAttribute((const void *)lds_fill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(lds_fill_kernel, dim3(256 * 2 * 8), dim3(256), 64 * 1024, 0, value, d_sink);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    return (int)hipDeviceSynchronize();
}
