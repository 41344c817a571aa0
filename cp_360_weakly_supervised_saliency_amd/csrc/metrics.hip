// K8: saliency metrics of utils/eval_saliency.py on the device: AUC_Judd (:90-146),
// CorrCoeff (:149-176), similarity (:179-190), AUC_Borji (:14-87), each behind the resize the reference applies
// first - cv2.resize(x, (240, 120), cv2.INTER_LANCZOS4) with the flag in the `dst` slot, i.e. OpenCV's default
// INTER_LINEAR (half-pixel centres, edge clamp).  Maps are tiny (120 x 240 after the resize): every metric is a
// few reductions plus, for the AUCs, a rank computation.  No sort is needed: the ROC point of the i-th largest
// fixated value only needs i (its rank among the fixated values) and the number of pixels at or above it, both
// plain counts - O(n_fix * n) compares, ~1e8 at most, spread over the chip.
// Arithmetic follows the reference's dtypes where they decide a comparison (float32 maps, float64 thresholds and
// jitter) and float64 for the sums (the reference's float32 pairwise sums differ from them by ~1e-7 relative).
#include "common.h"

// numpy / Pillow evaluate a * b + c as two rounded operations: no FMA contraction in this file (HIP's default
// -ffp-contract=fast would fuse them and change the last bit).  Plain operators only: HIP's __fadd_rn / __fmul_rn are header
// functions compiled with contraction ON, and their instructions keep that flag when inlined here.
#pragma clang fp contract(off)

namespace {

// cv2.resize INTER_LINEAR for one axis: source index / fraction of destination index d
__device__ __forceinline__ void lin_axis(int d, int n_src, int n_dst, int& i0, int& i1, float& fr) {
    const double scale = (double)n_src / (double)n_dst;
    const double f = ((double)d + 0.5) * scale - 0.5;
    int i = (int)floor(f);
    float t = (float)(f - (double)i);
    if (i < 0) { i = 0; t = 0.f; }
    if (i >= n_src - 1) { i = n_src - 1; t = 0.f; }
    i0 = i;
    i1 = min(i + 1, n_src - 1);
    fr = t;
}

__global__ __launch_bounds__(256) void resize_linear_kernel(const float* __restrict__ src, int h, int w,
                                                            float* __restrict__ dst, int dh, int dw) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= dh * dw) return;
    const int y = idx / dw, x = idx - y * dw;
    int y0, y1, x0, x1;
    float fy, fx;
    lin_axis(y, h, dh, y0, y1, fy);
    lin_axis(x, w, dw, x0, x1, fx);
    // same operation order as the oracle (and float32 throughout): rows first, then the vertical blend
    const float a = src[y0 * w + x0], b = src[y0 * w + x1], c = src[y1 * w + x0], d = src[y1 * w + x1];
    const float top = a * (1.f - fx) + b * fx;
    const float bot = c * (1.f - fx) + d * fx;
    dst[idx] = top * (1.f - fy) + bot * fy;
}

// ---- block-wide reductions (1024 threads)
template <typename V, typename Op>
__device__ __forceinline__ V block_reduce(V v, Op op, V* sm) {
    const int tid = threadIdx.x;
    sm[tid] = v;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (tid < s) sm[tid] = op(sm[tid], sm[tid + s]);
        __syncthreads();
    }
    const V r = sm[0];
    __syncthreads();
    return r;
}
struct OpAdd { __device__ double operator()(double a, double b) const { return a + b; } };
struct OpMin { __device__ double operator()(double a, double b) const { return a < b ? a : b; } };
struct OpMax { __device__ double operator()(double a, double b) const { return a > b ? a : b; } };

// Workspace layout (doubles): [0] n_fix, [1] unused, then S [n] (normalised saliency, f64), Sth [n] (values at
// fixated pixels, compacted), pts [2 * (n + 2)] (fp, tp by rank).
// mode 0 (AUC_Judd): S = sal (+ jitter); S = (S - min) / (max - min) - in float64 with the (float64) jitter, in float32
//   without it, as numpy's type promotion does                                                          :104-116
// mode 1 (AUC_Borji): S[S > mean + 2 std] = 1 (float32), then the same normalisation in float32         :37-40
__global__ __launch_bounds__(1024) void auc_prepare_kernel(const float* __restrict__ sal, const float* __restrict__ fix,
                                                           const double* __restrict__ jitter, int n, int mode,
                                                           double* __restrict__ work) {
    __shared__ double sm[1024];
    __shared__ int cnt;
    const int tid = threadIdx.x;
    double* S = work + 2;
    double* Sth = S + n;
    // fixation threshold mean(F) + 2 std(F) (:123 / :48)
    double s1 = 0.0;
    for (int i = tid; i < n; i += 1024) s1 += (double)fix[i];
    const double fmean = block_reduce(s1, OpAdd(), sm) / n;
    double s2 = 0.0;
    for (int i = tid; i < n; i += 1024) { const double d = (double)fix[i] - fmean; s2 += d * d; }
    const double fstd = sqrt(block_reduce(s2, OpAdd(), sm) / n);
    const float fthr = (float)(fmean + 2.0 * fstd);
    if (mode == 1) {
        double a1 = 0.0;
        for (int i = tid; i < n; i += 1024) a1 += (double)sal[i];
        const double smean = block_reduce(a1, OpAdd(), sm) / n;
        double a2 = 0.0;
        for (int i = tid; i < n; i += 1024) { const double d = (double)sal[i] - smean; a2 += d * d; }
        const float sthr = (float)(smean + 2.0 * sqrt(block_reduce(a2, OpAdd(), sm) / n));
        double mn = 1e300, mx = -1e300;
        for (int i = tid; i < n; i += 1024) {
            const float v = sal[i] > sthr ? 1.0f : sal[i];
            mn = fmin(mn, (double)v);
            mx = fmax(mx, (double)v);
        }
        const float fmn = (float)block_reduce(mn, OpMin(), sm), fmx = (float)block_reduce(mx, OpMax(), sm);
        for (int i = tid; i < n; i += 1024) {
            const float v = sal[i] > sthr ? 1.0f : sal[i];
            S[i] = (double)((v - fmn) / (fmx - fmn));
        }
    } else if (!jitter) {
        // jitter=False: the reference normalises the float32 map in float32 (no float64 randn term promotes it, :104-116)
        double mn = 1e300, mx = -1e300;
        for (int i = tid; i < n; i += 1024) {
            mn = fmin(mn, (double)sal[i]);
            mx = fmax(mx, (double)sal[i]);
        }
        const float fmn = (float)block_reduce(mn, OpMin(), sm), fmx = (float)block_reduce(mx, OpMax(), sm);
        for (int i = tid; i < n; i += 1024) S[i] = (double)((sal[i] - fmn) / (fmx - fmn));
    } else {
        double mn = 1e300, mx = -1e300;
        for (int i = tid; i < n; i += 1024) {
            const double v = (double)sal[i] + (jitter ? jitter[i] : 0.0);
            mn = fmin(mn, v);
            mx = fmax(mx, v);
        }
        const double gmn = block_reduce(mn, OpMin(), sm), gmx = block_reduce(mx, OpMax(), sm);
        for (int i = tid; i < n; i += 1024) {
            const double v = (double)sal[i] + (jitter ? jitter[i] : 0.0);
            S[i] = (v - gmn) / (gmx - gmn);
        }
    }
    if (tid == 0) cnt = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024)
        if (fix[i] > fthr) Sth[atomicAdd(&cnt, 1)] = S[i];       // order is irrelevant: ranks are recomputed
    __syncthreads();
    if (tid == 0) work[0] = (double)cnt;
}

// ROC point of every fixated value (:126-141): rank i among the fixated values (descending; ties by position),
// aboveth = #{S >= value};  tp = i / n_fix,  fp = (aboveth - i) / (n - n_fix), stored at index i + 1.
__global__ __launch_bounds__(256) void auc_judd_rank_kernel(int n, double* __restrict__ work) {
    const int n_fix = (int)work[0];
    const double* S = work + 2;
    const double* Sth = S + n;
    double* fp = const_cast<double*>(Sth) + n;
    double* tp = fp + (n + 2);
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j == 0) {
        fp[0] = 0.0; tp[0] = 0.0;
        fp[n_fix + 1] = 1.0; tp[n_fix + 1] = 1.0;
    }
    if (j >= n_fix) return;
    const double v = Sth[j];
    int rank = 0, above = 0;
    for (int k = 0; k < n_fix; ++k) rank += (Sth[k] > v) || (Sth[k] == v && k < j);
    for (int k = 0; k < n; ++k) above += S[k] >= v;
    tp[rank + 1] = (double)rank / (double)n_fix;
    fp[rank + 1] = (double)(above - rank) / (double)(n - n_fix);
}

// np.trapz(tp, fp) over n_fix + 2 points (:143)
__global__ __launch_bounds__(1024) void auc_trapz_kernel(int n, const double* __restrict__ work, double* __restrict__ out) {
    __shared__ double sm[1024];
    const int n_fix = (int)work[0];
    const double* fp = work + 2 + 2 * (size_t)n;
    const double* tp = fp + (n + 2);
    double s = 0.0;
    for (int k = threadIdx.x; k < n_fix + 1; k += 1024) s += (fp[k + 1] - fp[k]) * (tp[k + 1] + tp[k]) * 0.5;
    const double r = block_reduce(s, OpAdd(), sm);
    if (threadIdx.x == 0) { out[0] = r; out[1] = (double)n_fix; }
}

// AUC_Borji (:53-80): one workgroup per random split; rr int32 [n_fix, n_splits] pixel indices.
__global__ __launch_bounds__(256) void auc_borji_kernel(int n, int n_splits, const int* __restrict__ rr, double step,
                                                        const double* __restrict__ work, double* __restrict__ aucs) {
    __shared__ double sm[256];
    __shared__ double tpv[128], fpv[128];
    const int n_fix = (int)work[0];
    const double* S = work + 2;
    const double* Sth = S + n;
    const int ss = blockIdx.x, tid = threadIdx.x;
    double mx = -1e300;
    for (int k = tid; k < n_fix; k += 256) mx = fmax(mx, fmax(Sth[k], S[rr[(size_t)k * n_splits + ss]]));
    const double top = block_reduce(mx, OpMax(), sm);
    const int nthr = (int)ceil(top / step);                  // len(np.arange(0.0, top, step)); S in [0, 1]: <= 101
    // thresholds descending: t_i = (nthr - 1 - i) * step; tp[i + 1] = #(Sth >= t_i) / n_fix, fp likewise on curfix
    for (int i = 0; i < nthr; ++i) {
        const double t = (double)(nthr - 1 - i) * step;
        double a = 0.0, b = 0.0;
        for (int k = tid; k < n_fix; k += 256) {
            a += Sth[k] >= t;
            b += S[rr[(size_t)k * n_splits + ss]] >= t;
        }
        const double ta = block_reduce(a, OpAdd(), sm), tb = block_reduce(b, OpAdd(), sm);
        if (tid == 0) { tpv[i + 1] = ta / (double)n_fix; fpv[i + 1] = tb / (double)n_fix; }
    }
    if (tid == 0) {
        tpv[0] = 0.0; fpv[0] = 0.0;
        tpv[nthr + 1] = 1.0; fpv[nthr + 1] = 1.0;
        double s = 0.0;
        for (int k = 0; k < nthr + 1; ++k) s += (fpv[k + 1] - fpv[k]) * (tpv[k + 1] + tpv[k]) * 0.5;
        aucs[ss] = s;
    }
}

// CorrCoeff (:149-176) and similarity (:179-190) of two resized maps; out[0] = CC, out[1] = SIM
__global__ __launch_bounds__(1024) void cc_sim_kernel(const float* __restrict__ a, const float* __restrict__ b, int n,
                                                      double* __restrict__ out) {
    __shared__ double sm[1024];
    const int tid = threadIdx.x;
    double sa = 0, sb = 0, mna = 1e300, mxa = -1e300, mnb = 1e300, mxb = -1e300;
    for (int i = tid; i < n; i += 1024) {
        const double x = a[i], y = b[i];
        sa += x; sb += y;
        mna = fmin(mna, x); mxa = fmax(mxa, x);
        mnb = fmin(mnb, y); mxb = fmax(mxb, y);
    }
    const double ma = block_reduce(sa, OpAdd(), sm) / n, mb = block_reduce(sb, OpAdd(), sm) / n;
    mna = block_reduce(mna, OpMin(), sm); mxa = block_reduce(mxa, OpMax(), sm);
    mnb = block_reduce(mnb, OpMin(), sm); mxb = block_reduce(mxb, OpMax(), sm);
    double va = 0, vb = 0;
    for (int i = tid; i < n; i += 1024) {
        const double x = a[i] - ma, y = b[i] - mb;
        va += x * x; vb += y * y;
    }
    const double sda = sqrt(block_reduce(va, OpAdd(), sm) / n), sdb = sqrt(block_reduce(vb, OpAdd(), sm) / n);
    // standardised maps (:154-155), their means (:166-167, ~0) and the correlation sums (:169-174)
    double za = 0, zb = 0;
    for (int i = tid; i < n; i += 1024) { za += (a[i] - ma) / sda; zb += (b[i] - mb) / sdb; }
    const double am = block_reduce(za, OpAdd(), sm) / n, bm = block_reduce(zb, OpAdd(), sm) / n;
    double c = 0, d = 0, e = 0, na = 0, nb = 0;
    for (int i = tid; i < n; i += 1024) {
        const double x = (a[i] - ma) / sda - am, y = (b[i] - mb) / sdb - bm;
        c += x * y; d += x * x; e += y * y;
        na += (a[i] - mna) / (mxa - mna);
        nb += (b[i] - mnb) / (mxb - mnb);
    }
    c = block_reduce(c, OpAdd(), sm); d = block_reduce(d, OpAdd(), sm); e = block_reduce(e, OpAdd(), sm);
    na = block_reduce(na, OpAdd(), sm); nb = block_reduce(nb, OpAdd(), sm);
    double sim = 0;
    for (int i = tid; i < n; i += 1024)
        sim += fmin((a[i] - mna) / (mxa - mna) / na, (b[i] - mnb) / (mxb - mnb) / nb);
    sim = block_reduce(sim, OpAdd(), sm);
    if (tid == 0) { out[0] = c / sqrt(d * e); out[1] = sim; }
}

}  // namespace

extern "C" int cp360_resize_linear_f32(const float* src, int h, int w, float* dst, int dh, int dw, void* stream) {
    if (!src || !dst) return CP360_ERR_NULL;
    if (h <= 0 || w <= 0 || dh <= 0 || dw <= 0) return CP360_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(resize_linear_kernel, dim3((dh * dw + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, h, w,
                       dst, dh, dw);
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" size_t cp360_metric_work_bytes(int n) { return n > 0 ? sizeof(double) * (2 + 4 * (size_t)n + 4) : 0; }

extern "C" int cp360_metric_auc_prepare(const float* sal, const float* fix, const double* jitter, int n, int mode,
                                        void* work, void* stream) {
    if (!sal || !fix || !work) return CP360_ERR_NULL;
    if (n <= 0 || (mode != 0 && mode != 1)) return CP360_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(auc_prepare_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, sal, fix, jitter, n, mode,
                       (double*)work);
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" int cp360_metric_auc_judd(int n, void* work, double* out2, void* stream) {
    if (!work || !out2) return CP360_ERR_NULL;
    if (n <= 0) return CP360_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(auc_judd_rank_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, (double*)work);
    hipLaunchKernelGGL(auc_trapz_kernel, dim3(1), dim3(1024), 0, st, n, (const double*)work, out2);
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" int cp360_metric_auc_borji(int n, int n_splits, const int* rr, double step, void* work, double* aucs,
                                      void* stream) {
    if (!work || !rr || !aucs) return CP360_ERR_NULL;
    if (n <= 0 || n_splits <= 0 || !(step >= 0.0099)) return CP360_ERR_BAD_SHAPE;      // <= 101 thresholds (LDS table)
    hipLaunchKernelGGL(auc_borji_kernel, dim3(n_splits), dim3(256), 0, (hipStream_t)stream, n, n_splits, rr, step,
                       (const double*)work, aucs);
    CP360_CHECK_HIP();
    return CP360_OK;
}

extern "C" int cp360_metric_cc_sim(const float* a, const float* b, int n, double* out2, void* stream) {
    if (!a || !b || !out2) return CP360_ERR_NULL;
    if (n <= 0) return CP360_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(cc_sim_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, b, n, out2);
    CP360_CHECK_HIP();
    return CP360_OK;
}
