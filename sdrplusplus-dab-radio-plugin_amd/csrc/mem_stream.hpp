// mem_stream.hpp -- loads and stores for data a launch touches exactly once (IQ samples in, soft bits out, survivor
// words out and back in).  They carry the ISA's "nt" (non-temporal) bit: the line is not kept in L2 / the memory-side
// cache at the expense of lines that are still being assembled by partial writes.  Measured on the fused OFDM kernel,
// same buffers, launches alternated inside one process (tools/ab_inproc.py): 5.73 ms -> 5.51 ms per 16 384 frames.
#pragma once
#include <hip/hip_runtime.h>

namespace dabk {

namespace detail {
typedef float nt_f4 __attribute__((ext_vector_type(4)));
typedef float nt_f2 __attribute__((ext_vector_type(2)));
typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
typedef unsigned nt_u2 __attribute__((ext_vector_type(2)));
}  // namespace detail

__device__ __forceinline__ float4 ld_stream(const float4 *p) {
    const detail::nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const detail::nt_f4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float2 ld_stream(const float2 *p) {
    const detail::nt_f2 v = __builtin_nontemporal_load(reinterpret_cast<const detail::nt_f2 *>(p));
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ uint4 ld_stream(const uint4 *p) {
    const detail::nt_u4 v = __builtin_nontemporal_load(reinterpret_cast<const detail::nt_u4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 ld_stream(const uint2 *p) {
    const detail::nt_u2 v = __builtin_nontemporal_load(reinterpret_cast<const detail::nt_u2 *>(p));
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void st_stream(uint4 *p, const uint4 v) {
    __builtin_nontemporal_store(detail::nt_u4{v.x, v.y, v.z, v.w}, reinterpret_cast<detail::nt_u4 *>(p));
}
__device__ __forceinline__ void st_stream(uint2 *p, const uint2 v) {
    __builtin_nontemporal_store(detail::nt_u2{v.x, v.y}, reinterpret_cast<detail::nt_u2 *>(p));
}
__device__ __forceinline__ void st_stream(float2 *p, const float2 v) {
    __builtin_nontemporal_store(detail::nt_f2{v.x, v.y}, reinterpret_cast<detail::nt_f2 *>(p));
}
__device__ __forceinline__ void st_stream(float4 *p, const float4 v) {
    __builtin_nontemporal_store(detail::nt_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<detail::nt_f4 *>(p));
}

}  // namespace dabk
