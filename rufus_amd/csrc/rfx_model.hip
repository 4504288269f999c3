// Row N4: the coverage model fit of ModelDist (reference src/ModelDist.cpp), evaluated on the device.
//
// The reference's search is a chain of ~70 dependent steps; a step scores 11 candidate models, a score is a
// 10^4 x 360 table of normal densities (rows = histogram index, columns = copy number) that is normalised by
// column, turned into copy-number weights through the cells (SC*a, a+1), and compared with the histogram on
// [inflection, 5*SC).  Per step that is 4*10^7 exp() -- 0.2 ms of binary64 work on an MI355X, so the job is
// bound by the LATENCY of a step, not by any throughput roofline: four launches and one 88-byte read-back.
//   k_model_colsum   one workgroup per (candidate, column): sum of the column over all rows (fixed-shape tree)
//   k_model_weights  one wave per copy number: its weight from the cell (SC*a, a+1) and that row's total
//   k_model_terms    one wave per compared row: (histogram - model)^2 or (ln histogram - ln model)^2
//   k_model_sum      the residual of each candidate
//   k_model_dist / k_model_rowtot   the tables of the final model for the output files
// Compiled with -ffp-contract=off: a*b+c stays two roundings, as in the reference's x86-64 build.
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "../../include/rufus_hip.h"
#include "rfx_internal.h"

namespace {

constexpr double kPi = 3.14159;  // ModelDist.cpp:28 (sic)
constexpr int MD_BLOCK = 256;
constexpr int MD_MAX_CAND = 64;

struct md_cand {
  double sc, stdev, factor, skew, power;
  int jn;       // copies 1..jn (j < n / sc, ModelDist.cpp:95); columns 1..jn+1 with the half-copy column first
  int n_terms;  // rows compared with the histogram: i = inflection, i < sc * max_copy (a double loop variable, :183)
};

__device__ inline double md_norm(double x, double mu, double sigma, double skew, double p) {  // :31-37
  // (skew is 0 in every model the reference's own search makes -- its skew loop never runs -- and pow(+-0, p > 0) = 0)
  if (x < mu && !(skew == 0 && p > 0)) sigma = sigma + pow((mu - x) * skew, p);
  return (1 / (sqrt(2 * kPi * (sigma * sigma)))) * exp(-(((x - mu) * (x - mu)) / (2 * (sigma * sigma))));
}

__device__ inline void md_column(const md_cand& m, int c, double& mu, double& sigma) {  // :89-99
  if (c == 1) {
    mu = m.sc / 2;
    sigma = m.stdev * (1 - ((1 - (m.stdev / (m.stdev + (1 * m.factor)))) / 2));
  } else {
    const double j = (double)(c - 1);
    mu = m.sc * j;
    sigma = m.stdev + ((j - 1) * m.factor);
  }
}

// the table cell (row, c): columns 1..jn are divided by their sum over the rows, the last one is left raw (:107-117)
__device__ inline double md_cell(const md_cand& m, const double* __restrict__ colsum, long row, int c) {
  double mu, sigma;
  md_column(m, c, mu, sigma);
  const double v = md_norm((double)row, mu, sigma, m.skew, m.power);
  return c <= m.jn ? v / colsum[c] : v;
}

__global__ __launch_bounds__(MD_BLOCK) void k_model_colsum(const md_cand* __restrict__ cands, int n, int row0,
                                                            int stride, double* __restrict__ colsum) {
  __shared__ double s_part[MD_BLOCK];
  const md_cand m = cands[blockIdx.y];
  const int c = blockIdx.x + 1;
  if (c > m.jn) return;
  double mu, sigma;
  md_column(m, c, mu, sigma);
  // beyond 39 sigma exp() has underflowed to exactly 0 (39^2 / 2 > 745.2): only the band of the column is evaluated
  int lo = row0, hi = n;
  if (m.skew == 0 && m.power > 0 && sigma > 0 && sigma < 1e6) {
    lo = max(row0, (int)floor(mu - 39 * sigma));
    hi = min(n, (int)ceil(mu + 39 * sigma) + 1);
  }
  double acc = 0;
  for (int i = lo + threadIdx.x; i < hi; i += MD_BLOCK) acc += md_norm((double)i, mu, sigma, m.skew, m.power);
  s_part[threadIdx.x] = acc;
  __syncthreads();
  for (int w = MD_BLOCK / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) s_part[threadIdx.x] += s_part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) colsum[(size_t)blockIdx.y * stride + c] = s_part[0];
}

// testModel / testModelLog after the tables (:130-198, :262-318): one WAVE per table row -- a lane evaluates every
// 64th cell of the row, the wave adds the partial sums up through the lanes -- and four rows per workgroup, so that
// the ~500 rows a candidate needs are spread over the chip.  (Earlier versions, measured on the testRun table: one
// lane per row adding its 360 cells in the reference's column order, 595 us per step; the same with the cells of a row
// computed by a wave and read back from LDS by lane 0, 420 us -- 360 dependent LDS round trips; one workgroup per
// candidate with wave sums, 205 us -- 11 CUs busy with binary64 exp and divisions.  Sums of ~360 positive terms in
// another order differ by ~1e-16 relative; an infinity or a NaN among them comes out as itself in any order.)
__device__ inline double md_wave_sum(double v) {
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// copy-number weights RC (:130-163): column a + 1 from the cell (SC * a, a + 1) and the share of that column in its row
__global__ __launch_bounds__(MD_BLOCK) void k_model_weights(const md_cand* __restrict__ cands,
                                                             const long long* __restrict__ histo, int stride,
                                                             const double* __restrict__ colsum_all,
                                                             double* __restrict__ rc_all) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const md_cand m = cands[blockIdx.y];
  const double* colsum = colsum_all + (size_t)blockIdx.y * stride;
  double* rc = rc_all + (size_t)blockIdx.y * stride;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const long s = (long)m.sc, h = (long)(m.sc / 2);
    const double at_s = md_cell(m, colsum, s, 2);
    const double t_sc = (double)histo[s] / at_s;
    const double het = ((double)histo[h] - (md_cell(m, colsum, h, 2) * t_sc)) / md_cell(m, colsum, h, 1);
    rc[0] = 0;
    rc[1] = het > 0 ? het : 0;
    rc[2] = (double)histo[s] / at_s;
  }
  const int a = 2 + (int)blockIdx.x * (MD_BLOCK / 64) + wave;
  if (a > m.jn) return;
  const long r = (long)(m.sc * a);
  double part = 0;
  for (int c = 1 + lane; c <= m.jn; c += 64) part += md_cell(m, colsum, r, c);
  const double total = md_wave_sum(part);
  if (lane == 0) {
    const double d = md_cell(m, colsum, r, a + 1);
    rc[a + 1] = (double)histo[r] / d * (d / total);
  }
}

// squared differences between histogram and model on the compared rows (:176-195)
__global__ __launch_bounds__(MD_BLOCK) void k_model_terms(const md_cand* __restrict__ cands,
                                                           const long long* __restrict__ histo, int n, int stride,
                                                           const double* __restrict__ colsum_all,
                                                           const double* __restrict__ rc_all, int log_resid,
                                                           int inflection, double* __restrict__ terms) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const md_cand m = cands[blockIdx.y];
  const int t = (int)blockIdx.x * (MD_BLOCK / 64) + wave;
  if (t >= m.n_terms) return;
  const double* colsum = colsum_all + (size_t)blockIdx.y * stride;
  const double* rc = rc_all + (size_t)blockIdx.y * stride;
  const long row = (long)inflection + t;
  double part = 0;
  for (int c = 1 + lane; c <= m.jn; c += 64) part += md_cell(m, colsum, row, c) * rc[c];
  const double sum = md_wave_sum(part);
  if (lane == 0) {
    const double hv = (double)histo[row];
    const double dlt = log_resid ? log(hv) - log(sum) : hv - sum;
    terms[(size_t)blockIdx.y * n + t] = dlt * dlt;
  }
}

__global__ __launch_bounds__(MD_BLOCK) void k_model_sum(const md_cand* __restrict__ cands, int n,
                                                         const double* __restrict__ terms, double* __restrict__ resid) {
  __shared__ double s_part[MD_BLOCK / 64];
  const int n_terms = cands[blockIdx.x].n_terms;
  double acc = 0;
  for (int t = threadIdx.x; t < n_terms; t += MD_BLOCK) acc += terms[(size_t)blockIdx.x * n + t];
  acc = md_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) resid[blockIdx.x] = ((s_part[0] + s_part[1]) + s_part[2]) + s_part[3];
}

__global__ __launch_bounds__(MD_BLOCK) void k_model_dist(const md_cand* __restrict__ cands, int n, int n_cols,
                                                          const double* __restrict__ colsum, double* __restrict__ dist) {
  const md_cand m = cands[0];
  const size_t total = (size_t)n * (n_cols + 1);
  for (size_t at = (size_t)blockIdx.x * MD_BLOCK + threadIdx.x; at < total; at += (size_t)gridDim.x * MD_BLOCK) {
    const long row = (long)(at / (n_cols + 1));
    const int c = (int)(at % (n_cols + 1));
    dist[at] = c == 0 ? 0.0 : md_cell(m, colsum, row, c);
  }
}

__global__ __launch_bounds__(MD_BLOCK) void k_model_rowtot(int n, int n_cols, int jn, const double* __restrict__ dist,
                                                            double* __restrict__ rowtot) {
  const int row = blockIdx.x * MD_BLOCK + threadIdx.x;
  if (row >= n) return;
  double total = 0;
  for (int c = 1; c <= jn; ++c) total += dist[(size_t)row * (n_cols + 1) + c];  // :764-766, in that order
  rowtot[row] = total;
}

// copies j = 1, 2, ... while j < n / sc (a size_t divided by a double, :95)
bool md_prepare(const rfx_model_params& p, uint32_t n, md_cand& out) {
  if (!(p.sc > 0) || !std::isfinite(p.sc)) return false;
  const double q = (double)n / p.sc;
  if (!(q < 1e6)) return false;
  int jn = 0;
  for (long j = 1; (double)j < q; ++j) ++jn;
  out = md_cand{p.sc, p.stdev, p.factor, p.skew, p.power, jn, 0};
  return true;
}

struct md_buffers {
  rfx_ctx* c;
  std::vector<void*> all;
  explicit md_buffers(rfx_ctx* ctx) : c(ctx) {}
  template <class T>
  T* get(size_t count) {
    void* p = rfxi::dmalloc(c, count * sizeof(T) + 8);
    if (p) all.push_back(p);
    return (T*)p;
  }
  ~md_buffers() {
    for (void* p : all) rfxi::dfree(c, p);
  }
};

int md_fail(hipError_t e, const char* what) {
  std::string msg = std::string(what) + ": " + hipGetErrorString(e);
  rfxi::set_error(msg.c_str());
  return RFX_E_HIP;
}

}  // namespace

extern "C" {

int rfx_model_residuals(rfx_ctx* c, const int64_t* histo, uint32_t n, const rfx_model_params* cand, int n_cand,
                        int log_resid, int inflection, int max_copy, double* resid_out) {
  if (!c || !histo || !cand || !resid_out || n < 4 || n_cand < 1 || n_cand > MD_MAX_CAND || inflection < 1 ||
      max_copy < 1)
    return RFX_E_INVAL;
  (void)hipSetDevice(c->device);
  std::vector<md_cand> m((size_t)n_cand);
  int jn_max = 0, terms_max = 0;
  for (int b = 0; b < n_cand; ++b) {
    if (!md_prepare(cand[b], n, m[(size_t)b])) return RFX_E_INVAL;
    // cells the reference reads: (SC, 2), (SC/2, 1), (SC*a, a+1), rows [inflection, SC*max_copy) -- inside the table?
    if ((long)(cand[b].sc / 2) < 1 || m[(size_t)b].jn < 2) return RFX_E_INVAL;
    if (!(cand[b].sc * max_copy <= (double)n) || (double)inflection >= (double)n) return RFX_E_RANGE;
    jn_max = std::max(jn_max, m[(size_t)b].jn);
    for (double i = inflection; i < cand[b].sc * max_copy; i++) ++m[(size_t)b].n_terms;
    terms_max = std::max(terms_max, m[(size_t)b].n_terms);
  }
  const int stride = jn_max + 2;
  md_buffers buf(c);
  md_cand* d_c = buf.get<md_cand>((size_t)n_cand);
  long long* d_h = buf.get<long long>(n);
  double* d_cs = buf.get<double>((size_t)n_cand * stride);
  double* d_rc = buf.get<double>((size_t)n_cand * stride);
  double* d_terms = buf.get<double>((size_t)n_cand * n);
  double* d_res = buf.get<double>((size_t)n_cand);
  if (!d_c || !d_h || !d_cs || !d_rc || !d_terms || !d_res) return RFX_E_NOMEM;
  hipError_t e = hipMemcpyAsync(d_c, m.data(), m.size() * sizeof(md_cand), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_h, histo, (size_t)n * 8, hipMemcpyHostToDevice, c->stream);
  if (e != hipSuccess) return md_fail(e, "rfx_model_residuals");
  {
    rfx_span sp(c, "k_model_colsum");
    hipLaunchKernelGGL(k_model_colsum, dim3((unsigned)jn_max, (unsigned)n_cand), dim3(MD_BLOCK), 0, c->stream, d_c, (int)n,
                       1, stride, d_cs);
  }
  constexpr unsigned rows_per_block = MD_BLOCK / 64;
  {
    rfx_span sp(c, "k_model_weights");
    hipLaunchKernelGGL(k_model_weights, dim3(((unsigned)jn_max - 1 + rows_per_block - 1) / rows_per_block, (unsigned)n_cand),
                       dim3(MD_BLOCK), 0, c->stream, d_c, d_h, stride, d_cs, d_rc);
  }
  if (terms_max > 0) {
    rfx_span sp(c, "k_model_terms");
    hipLaunchKernelGGL(k_model_terms, dim3(((unsigned)terms_max + rows_per_block - 1) / rows_per_block, (unsigned)n_cand),
                       dim3(MD_BLOCK), 0, c->stream, d_c, d_h, (int)n, stride, d_cs, d_rc, log_resid, inflection, d_terms);
  }
  {
    rfx_span sp(c, "k_model_sum");
    hipLaunchKernelGGL(k_model_sum, dim3((unsigned)n_cand), dim3(MD_BLOCK), 0, c->stream, d_c, (int)n, d_terms, d_res);
  }
  e = rfxi::queue_read(c, resid_out, d_res, (size_t)n_cand * 8);
  if (e == hipSuccess) e = rfxi::sync(c);
  return e == hipSuccess ? RFX_OK : md_fail(e, "rfx_model_residuals");
}

int rfx_model_tables(rfx_ctx* c, uint32_t n, const rfx_model_params* model, uint32_t* n_cols_out, double* dist,
                     size_t dist_cap, double* rowtot) {
  if (!c || !model || !n_cols_out || n < 4) return RFX_E_INVAL;
  (void)hipSetDevice(c->device);
  md_cand m;
  if (!md_prepare(*model, n, m) || m.jn < 1) return RFX_E_INVAL;
  const int n_cols = m.jn + 1;
  *n_cols_out = (uint32_t)n_cols;
  const size_t cells = (size_t)n * (n_cols + 1);
  if (!dist || !rowtot || dist_cap < cells) return RFX_E_RANGE;  // (n_cols_out is set: the caller can size and repeat)
  md_buffers buf(c);
  md_cand* d_c = buf.get<md_cand>(1);
  double* d_cs = buf.get<double>((size_t)n_cols + 2);
  double* d_dist = buf.get<double>(cells);
  double* d_tot = buf.get<double>(n);
  if (!d_c || !d_cs || !d_dist || !d_tot) return RFX_E_NOMEM;
  hipError_t e = hipMemcpyAsync(d_c, &m, sizeof m, hipMemcpyHostToDevice, c->stream);
  if (e != hipSuccess) return md_fail(e, "rfx_model_tables");
  e = hipStreamSynchronize(c->stream);  // (m is on this stack frame)
  if (e != hipSuccess) return md_fail(e, "rfx_model_tables");
  {
    rfx_span sp(c, "k_model_colsum");  // main() sums the columns from row 0 (:745-754), the search from row 1
    hipLaunchKernelGGL(k_model_colsum, dim3((unsigned)m.jn, 1), dim3(MD_BLOCK), 0, c->stream, d_c, (int)n, 0, n_cols + 2,
                       d_cs);
  }
  {
    rfx_span sp(c, "k_model_dist");
    hipLaunchKernelGGL(k_model_dist, dim3((unsigned)std::min<size_t>((cells + MD_BLOCK - 1) / MD_BLOCK, (size_t)c->n_cu * 16)),
                       dim3(MD_BLOCK), 0, c->stream, d_c, (int)n, n_cols, d_cs, d_dist);
    hipLaunchKernelGGL(k_model_rowtot, dim3((n + MD_BLOCK - 1) / MD_BLOCK), dim3(MD_BLOCK), 0, c->stream, (int)n, n_cols,
                       m.jn, d_dist, d_tot);
  }
  e = hipMemcpyAsync(dist, d_dist, cells * 8, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(rowtot, d_tot, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = rfxi::sync(c);
  return e == hipSuccess ? RFX_OK : md_fail(e, "rfx_model_tables");
}

}  // extern "C"
