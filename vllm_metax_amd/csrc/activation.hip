// activation.hip — SwiGLU gate: out[t, i] = T(silu(x)) * y, x = in[t, i], y = in[t, d+i].
//
// Reference restated: csrc/activation_kernels.cu:14-36 (compute / kernel),
// :142-147 (silu_kernel: T( x / (1 + expf(-x)) ) in fp32, rounded to T, then the
// T*T product is rounded again), :177-247 (launch).  2-D grid over (column tile,
// token) so that small decode batches still fill the chip; 16 B per lane.
#include "common.cuh"

namespace mi355x {

template <typename T, bool VEC>
__global__ void silu_and_mul_kernel(T* __restrict__ out, const T* __restrict__ in, int d) {
  const int64_t token = blockIdx.y;
  const T* x = in + token * 2 * d;
  const T* y = x + d;
  T* o = out + token * d;
  if constexpr (VEC) {
    constexpr int V = 16 / sizeof(T);
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (i >= d) return;
    const Vec16<T> xv = load16(x + i);
    const Vec16<T> yv = load16(y + i);
    Vec16<T> r;
#pragma unroll
    for (int j = 0; j < V; ++j) r.e[j] = mul_t<T>(silu_t<T>(xv.e[j]), yv.e[j]);
    store16(o + i, r);
  } else {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d) return;
    o[i] = mul_t<T>(silu_t<T>(x[i]), y[i]);
  }
}

}  // namespace mi355x

using namespace mi355x;

extern "C" int mi355x_silu_and_mul(void* out, const void* input, int num_tokens, int d,
                                   int dtype, mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && d > 0, MI355X_EINVAL, "silu_and_mul: bad sizes");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(out && input, MI355X_EINVAL, "silu_and_mul: null pointer");
  MI355X_REQUIRE(num_tokens <= 65535, MI355X_EUNSUPPORTED,
                 "silu_and_mul: num_tokens %d > 65535", num_tokens);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return MI355X_DISPATCH_FLOAT(dtype, [&] {
    constexpr int V = 16 / sizeof(scalar_t);
    scalar_t* o = static_cast<scalar_t*>(out);
    const scalar_t* in = static_cast<const scalar_t*>(input);
    const bool vec = d % V == 0 && (reinterpret_cast<uintptr_t>(o) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    const int work = vec ? d / V : d;
    const int threads = work >= 256 ? 256 : ((work + 63) / 64) * 64;
    dim3 grid((work + threads - 1) / threads, num_tokens), block(threads);
    if (vec)
      hipLaunchKernelGGL((silu_and_mul_kernel<scalar_t, true>), grid, block, 0, s, o, in, d);
    else
      hipLaunchKernelGGL((silu_and_mul_kernel<scalar_t, false>), grid, block, 0, s, o, in, d);
    return check_launch("silu_and_mul");
  });
}
