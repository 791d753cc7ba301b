// Row-sparse Adam for a small key list against a table of any size in one launch, without a sort: the algorithm is in adam_rows_owner.h.
#include "adam_rows_owner.h"

template <int VPT, int HS, class KeyT>
__global__ __launch_bounds__(SA_NT) void sparse_adam_owner_k(SaParams P) {
    extern __shared__ __align__(16) unsigned char sa_lds[];
    sa_body<VPT, HS, KeyT>(P, sa_lds);
}

extern "C" int re_sparse_adam_rows_small(const float* g, const void* keys, int32_t key_bytes, int32_t n_regions, int64_t region_stride,
                                         const int32_t* n_dev, int64_t n_mul, int64_t n_host, int64_t D, int64_t R, int64_t padding_idx,
                                         float* W, float* m, float* v, const float* hyper, int64_t step, double lr, double beta1,
                                         double beta2, double eps, double weight_decay, re_stream_t stream) {
    re_clear_error();
    if (!W || !m || !v || R <= 0 || n_regions < 0 || region_stride < 0 || n_host < 0 || n_mul < 0) return RE_EINVAL;
    if (D != 64 && D != 128) return RE_EUNSUPPORTED;
    if (key_bytes != 4 && key_bytes != 8) return RE_EINVAL;
    if (R >= 0xFFFFFFFEll || (int64_t)n_regions * region_stride >= 0xFFFFFFFFll) return RE_EUNSUPPORTED;
    if (!hyper && step < 1) return RE_EINVAL;
    if (n_regions == 0 || region_stride == 0 || (!n_dev && n_host == 0)) return RE_OK;
    if (!g || !keys) return RE_EINVAL;
    if (n_host > region_stride) return RE_EINVAL;
    if ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 7u)
        return RE_EINVAL;
    SaParams P;
    P.g = g; P.keys = keys; P.n_regions = n_regions; P.region_stride = region_stride; P.n_dev = n_dev; P.n_mul = n_mul; P.n_host = n_host;
    P.R = R; P.padding_idx = padding_idx; P.W = W; P.m = m; P.v = v;
    P.b1 = (float)beta1; P.b2 = (float)beta2; P.omb1 = (float)(1.0 - beta1); P.omb2 = (float)(1.0 - beta2);
    P.eps = (float)eps; P.wd = (float)weight_decay; P.hyper = hyper;
    P.step_size = 0.f; P.inv_sqrt_bc2 = 0.f;
    if (!hyper) {
        P.step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
        P.inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow(beta2, (double)step)));
    }
    hipStream_t s = (hipStream_t)stream;
    P.stride = D; P.coff = 0;
    if (D == 64) {
        if (key_bytes == 4) hipLaunchKernelGGL((sparse_adam_owner_k<1, 1, int32_t>), dim3(SA_NWG), dim3(SA_NT), SA_LDS_BYTES(1), s, P);
        else hipLaunchKernelGGL((sparse_adam_owner_k<1, 1, int64_t>), dim3(SA_NWG), dim3(SA_NT), SA_LDS_BYTES(1), s, P);
    } else {   // two 64-column pieces per row
        if (key_bytes == 4) hipLaunchKernelGGL((sparse_adam_owner_k<1, 2, int32_t>), dim3(SA_NWG), dim3(SA_NT), SA_LDS_BYTES(1), s, P);
        else hipLaunchKernelGGL((sparse_adam_owner_k<1, 2, int64_t>), dim3(SA_NWG), dim3(SA_NT), SA_LDS_BYTES(1), s, P);
    }
    return re_launch_status();
}
