// api.hip -- C ABI of libnabo_knn.so (include/nabo_knn.h).  Host orchestration only:
// buffer management, kernel sequencing on the index's HIP stream, HIP-event timing.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/nabo_knn.h"
#include "knn_common.h"

namespace nabo {
// kernels (pack.hip, l2_topk.hip, refine.hip, canberra.hip)
hipError_t centre_launch(const double *Y, int64_t n, int g, double *centre, hipStream_t st);
hipError_t pack_ref_launch(const double *Y, int64_t n, int g, const double *centre, double scale, int ksteps, int64_t ntiles_total,
                           const uint8_t *mask, float *out, unsigned int *norm_max_bits, hipStream_t st);
hipError_t pack_query_launch(const double *X, int64_t m, int g, const double *centre, double scale, int ksteps,
                             int64_t ntiles_total, float *out, double *xnorm, hipStream_t st);
hipError_t l2_topk_launch(int ksteps, int epl, const float *Xpk, const float *Ypk, int tiles_per_split, int S, int gx,
                          int64_t tile_off, int lkeep, uint32_t *cand_idx, float *cand_key, float *cand_tau,
                          hipStream_t st);
void l2_topk_geometry(int ksteps, int epl, int *rows_per_wg, int *wg_per_cu, int *lkeep_max);
// f16 operands of the matrix-pipe filters (pack.hip): K-concatenated tiles, nseg = 3 the f16x3 split, 1 the one-product form
hipError_t maxabs_launch(const double *V, int64_t n, int g, const double *centre, unsigned long long *out_bits, hipStream_t st);
hipError_t pack_cref_launch(const double *Y, int64_t n, int g, const double *centre, double scale, int kc,
                            int64_t ntiles_total, const uint8_t *mask, unsigned char *out, unsigned int *norm_max_bits,
                            bool layout16, hipStream_t st, const uint32_t *perm = nullptr, int nseg = 3);
hipError_t pack_cquery_launch(const double *X, int64_t m, int g, const double *centre, double scale, int kc,
                              int64_t ntiles_total, unsigned char *out, double *xnorm, bool layout16, hipStream_t st,
                              const uint32_t *perm = nullptr, int nseg = 3);
int l2q_pick_kc(int g);
int l2q_pick_kc1(int g);
// the one-product first pass (l2c_topk.hip; operands packed with layout16, nseg = 1)
hipError_t l2c_topk_launch(int kc, int geo, const unsigned char *Xpk, const unsigned char *Ypk, int tiles_per_split, int S,
                           int gx, int64_t tile_off, int lkeep, uint32_t *cand_idx, float *cand_key, float *cand_tau,
                           int64_t pad_tile, hipStream_t st, int64_t rows_valid, const float *tau_init, int tau_stride = 0,
                           int64_t tau_row0 = 0, const L2cPieces *pieces = nullptr);
// tournament seeds for the one-product pass (l2c_topk.hip: l2c_pre_kernel)
void l2c_pre_plan(int kc, int lkeep, int tiles_per_split, int scale_pct, int *pre_tiles, int *gt);
hipError_t l2c_pre_launch(int kc, int lkeep, const unsigned char *Xpk, const unsigned char *Ypk, int tiles_per_split, int S,
                          int64_t rows, int64_t tile_off, int pre_tiles, int gt, int64_t pad_tile, hipStream_t st,
                          int64_t rows_valid, float *tau_out, const int *ranges = nullptr, int rows_per_col = 0);
hipError_t merge_lists_launch(const uint32_t *cand_idx, const float *cand_key, const float *cand_tau, int64_t rows, int S, int L,
                              int lkeep, int Lout, uint32_t *out_idx, float *out_tau, hipStream_t st);
int l2c_pick_kc(int g);
int l2c_geometry(int kc, int lkeep_want, int pin);
void l2c_topk_geometry(int kc, int lkeep_want, int pin, int *rows_per_wg, int *wg_per_cu, int *lkeep_max);
// the same filter on v_mfma_f32_16x16x32_f16 (l2q_topk.hip; operands packed with layout16)
hipError_t l2q_topk_launch(int kc, const unsigned char *Xpk, const unsigned char *Ypk, int tiles_per_split, int S,
                           int gx, int64_t tile_off, int lkeep, uint32_t *cand_idx, float *cand_key, float *cand_tau,
                           int64_t pad_tile, hipStream_t st, const int32_t *wave_start = nullptr);
#ifdef NABO_EXPERIMENTS
// kernels of the experiments build only (tools/ab; measured slower than the product's, kept for A/B runs -- DESIGN.md 4.1b-d):
// the f16x3 split on v_mfma_f32_32x32x16_f16 (l2h_topk.hip), the same with reference tiles shared through an LDS ring
// (l2s_topk.hip), locality order of the streamed cells (order.hip)
hipError_t l2h_topk_launch(int kc, const unsigned char *Xpk, const unsigned char *Ypk, int tiles_per_split, int S,
                           int gx, int64_t tile_off, int lkeep, uint32_t *cand_idx, float *cand_key, float *cand_tau,
                           int64_t pad_tile, hipStream_t st);
void l2h_topk_geometry(int kc, int *rows_per_wg, int *wg_per_cu, int *lkeep_max);
int loc_key_bits(int g);
hipError_t loc_sort_temp_bytes(int64_t n, int nb, size_t *bytes);
hipError_t loc_order_launch(const double *V, int64_t n, int g, const double *centre, uint32_t *keys_a, uint32_t *pos_a,
                            uint32_t *keys_sorted, uint32_t *perm, void *temp, size_t temp_bytes, hipStream_t st);
hipError_t wave_start_launch(const uint32_t *tkeys, int64_t m, int rows_per_wave, const uint32_t *rkeys, int64_t n,
                             int64_t n_waves, int32_t *start, hipStream_t st);
hipError_t l2s_topk_launch(int kc, const unsigned char *Xpk, const unsigned char *Ypk, int tiles_per_split, int S, int gx,
                           int64_t tile_off, int lkeep, uint32_t *cand_idx, float *cand_key, float *cand_tau, hipStream_t st);
void l2s_topk_geometry(int kc, int *rows_per_wg, int *wg_per_cu, int *lkeep_max);
int l2s_pick_kc(int g);
#endif
void l2q_topk_geometry(int kc, int *rows_per_wg, int *wg_per_cu, int *lkeep_max);
hipError_t pairwise_launch(const double *X, int64_t m, const double *Y, int64_t n, int g, int metric, double f,
                           double *D, hipStream_t st);
hipError_t refine_launch(const double *X, int64_t row0, int64_t m, const double *Y, int g, const uint32_t *cand_idx,
                         const float *cand_tau, int S, int L, const double *xnorm, double err_coef, double ymax_sqrt,
                         double tau_scale, int k, int drop, int64_t base, int64_t n_valid_total, const uint32_t *masked_list,
                         int n_masked_list, int64_t *out_idx, double *out_dist, uint32_t *fail_rows,
                         unsigned int *fail_count, hipStream_t st, int metric = 0, double cb_f = 0.0,
                         float cb_plateau = 0.0f, int lvalid = 0, const uint32_t *rperm = nullptr,
                         const uint32_t *tperm = nullptr, float *fail_seed = nullptr);
hipError_t refine_cand_launch(const double *X, int64_t row0, int64_t m, const double *Y, int g, const uint32_t *cand_idx,
                              const float *cand_tau, int S, int L, const double *xnorm, double err_coef,
                              double ymax_sqrt, double tau_scale, int kout, int64_t base, int64_t n_valid_total,
                              int64_t *out_idx, double *out_dist, double *out_bound, hipStream_t st, int metric = 0,
                              int lvalid = 0, const uint32_t *rperm = nullptr, const uint32_t *tperm = nullptr);
hipError_t normalise_rows_launch(const double *X, int64_t m, int g, double *out, hipStream_t st);
hipError_t exact_rows_launch(const double *X, const double *Y, int64_t n, int g, int metric, double f,
                             const uint8_t *mask, const uint32_t *rows, unsigned int nrows, int k, int drop,
                             int64_t base, const uint32_t *masked_list, int n_masked_list, int64_t *out_idx,
                             double *out_dist, double *D, unsigned int d_rows, hipStream_t st);
hipError_t masked_tail_launch(const double *X, int64_t m, const double *Y, int g, int metric, double f,
                              const uint32_t *masked_list, int n_masked_list, int n_valid, int k, int drop,
                              int64_t base, int64_t *out_idx, double *out_dist, hipStream_t st);
hipError_t transpose_ref_launch(const double *Y, int64_t n, int g, double *Yt, hipStream_t st);
hipError_t canberra_topk_launch(int epl, const double *X, int64_t m, const double *Yt, int64_t n, int g, double f,
                                const uint8_t *mask, int S, double *cand_d, uint32_t *cand_i, hipStream_t st);
hipError_t merge_local_launch(const double *cand_d, const uint32_t *cand_i, int64_t m, int P, int k, int drop,
                              int64_t base, int64_t *out_idx, double *out_dist, int *n_found, hipStream_t st);
hipError_t merge_parts_launch(const double *parts_d, const int64_t *parts_i, int n_parts, int64_t m, int kp, int k,
                              int drop, int64_t *out_idx, double *out_dist, hipStream_t st);
// fp32 lower-bound Canberra filter (canberra_f32.hip) + helpers (refine.hip)
hipError_t cbf_pack_targets_launch(const double *X, int64_t m, int g, int gp, double f, float *xq, unsigned int *flag,
                                   hipStream_t st);
hipError_t cbf_pack_refs_launch(const double *Y, int64_t n, int g, int gp, float *ycf, unsigned int *flag,
                                hipStream_t st);
int cbf_pick_gp(int g);
void cbf_constants(int g, float *slack, float *plateau);
int cbf_lists_per_split();
int cbf_rows_per_wg(int epl);
hipError_t cbf_filter_launch(int gp, int epl, const float *xq, const void *xh, int64_t m, const float *ycf,
                             const void *ych, int64_t n, int g, const uint8_t *mask, int S, uint32_t *cand_idx,
                             float *cand_tau, hipStream_t st);
hipError_t cbf_pack_refs_rows_launch(const double *Y, int64_t n, int g, int gp, float *yrow, hipStream_t st);
hipError_t cbf_colminmax_launch(const double *Y, int64_t n, int g, unsigned int *colmm, hipStream_t st);
hipError_t cbf_pack_refs8_launch(const double *Y, int64_t n, int g, int gp, const double *quant, void *ych, hipStream_t st);
hipError_t cbf_pack_targets8_launch(const double *X, int64_t m, int g, int gp, double f, const double *quant, void *xh,
                                    hipStream_t st);
// the counting pass on per-bucket bitmaps (canberra_bits.hip)
int cbb_buckets();
int cbb_rows_per_wg();
bool cbb_available(int g, int gp, int epl);
size_t cbb_table_bytes(int64_t n, int g);
size_t cbb_valid_bytes(int64_t n);
hipError_t cbb_pack_table_launch(const double *Y, int64_t n, int g, const double *edges, uint32_t *tab, hipStream_t st);
hipError_t cbb_valid_launch(const uint8_t *mask, int64_t n, uint32_t *vbits, hipStream_t st);
hipError_t cbb_pack_targets_launch(const double *X, int64_t m, int g, int gp, double f, const double *edges, uint16_t *rowoff,
                                   hipStream_t st);
hipError_t cbb_filter_launch(int gp, const float *xq, const uint16_t *rowoff, int64_t m, const float *yrow, const uint32_t *tab,
                             const uint32_t *vbits, int64_t n, int g, int S, uint32_t *cand_idx, float *cand_tau, hipStream_t st);
hipError_t gather_rows_launch(const double *X, const uint32_t *rows, int64_t nrows, int g, double *out, hipStream_t st);
hipError_t iota_launch(uint32_t *out, int64_t n, hipStream_t st);
hipError_t scatter_rows_launch(const int64_t *si, const double *sd, const uint32_t *rows, int64_t nrows, int k,
                               int64_t *out_idx, double *out_dist, hipStream_t st);
hipError_t null_hist_launch(int64_t n_t, int P, uint64_t seed, int key_bits, const uint64_t *prefix, int done_bits,
                            unsigned int *hist, hipStream_t st);
hipError_t null_label_launch(int64_t n_t, int P, int W, uint64_t seed, int key_bits, const uint64_t *thr,
                             const uint8_t *group, uint32_t *bits, hipStream_t st);
hipError_t null_score_launch(int64_t n_ref, const int64_t *row_ptr, const int64_t *edge_t, const double *edge_w, int P,
                             int W, const uint32_t *bits, const int64_t *n_lab, int64_t n_a, double mult,
                             double *out_obs, int64_t *out_nge, double *out_mean, double *out_sd, hipStream_t st);
hipError_t snn_counts_launch(const int64_t *t_idx, int64_t m, const int64_t *r_idx, int64_t n, int k, int32_t *out,
                             hipStream_t st);
hipError_t csr_sort_temp_bytes(int64_t E, int64_t n_ref, size_t *bytes);
hipError_t csr_build_launch(const int64_t *edge_r, const int64_t *edge_t, const double *edge_w, int64_t E, int64_t n_ref,
                            int64_t n_t, uint32_t *keys_a, uint32_t *pos_a, uint32_t *keys_b, uint32_t *pos_b, void *temp,
                            size_t temp_bytes, int64_t *row_ptr, int64_t *out_t, double *out_w, unsigned int *flag,
                            hipStream_t st);
}  // namespace nabo

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) {                                                               \
            (void)hipGetLastError(); /* (the thread's sticky copy: a later launch check must not report THIS failure) */ \
            return fail(e__ == hipErrorOutOfMemory ? NABO_E_NOMEM : NABO_E_HIP, "%s failed: %s", \
                        #expr, hipGetErrorString(e__));                                        \
        }                                                                                      \
    } while (0)

int use_device(int device)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0)
        return fail(NABO_E_NODEVICE, "no HIP device is available (libnabo_knn has no CPU fallback)");
    if (device < 0 || device >= cnt) return fail(NABO_E_NODEVICE, "device %d out of range (have %d)", device, cnt);
    HIP_TRY(hipSetDevice(device));
    return NABO_OK;
}

// grow-only device buffer; owns its allocation (freed by release() or with the object: `delete ix` cannot miss a member)
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return NABO_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(NABO_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        }
        cap = want;
        return NABO_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

int pick_ksteps(int g)
{
    const int need = (g + 1) / 2;
    const int inst[] = {8, 16, 25, 32, 50, 64};
    for (int v : inst)
        if (need <= v) return v;
    return -1;
}

// Tuning options of an index (nabo_index_set_option; the defaults are the product's behaviour).  EVERY setting returns the
// same bits -- an option chooses how a launch is cut or which filter pass answers a row, never what the answer is.  The
// library reads two environment variables, once, in nabo_index_create: NABO_L2_MODE and NABO_CANBERRA_MODE (which first
// filter); -DNABO_EXPERIMENTS builds (tools/ab) also take every option below as NABO_OPT_<NAME>.
struct Options {
    int splits = 0;            // reference splits of a filter launch (0: the cost model decides)
    int tail_split = 1;        // the last, partially filled round of workgroups gets its own split count
    int lkeep = 0;             // kept entries of the first pass's lists (0: k' + 8)
    int coarse_slack = -1;     // kept entries of the one-product pass beyond k' + 8 (-1: 0 on 32-entry lists, 6 on 64-entry lists)
    int cand_slack = -1;       // candidate mode on the one-product pass: kept entries beyond the emitted ones (-1: the sharded query's rule)
    int seeded_pass = 1;       // links of the pass chain: rows the first pass fails go through the seeded one-product pass,
    int coarse_adapt = 1;      //   a weak one-product bound is remembered until the references change,
    int wide_retry = 1;        //   rows the 32-entry lists fail get 64-entry lists before the exact kernels
    int refine_overlap = 1;    // the refine of the main launch's rows runs beside the filter's tail launch
    int prepass = 100;         // tournament seeds: percent of the planned length (0: lists start from +inf)
    int pieces = 0;            // (-DNABO_EXPERIMENTS builds) fewer column-workgroups than slots: the launch cut into equal pieces of (column, tile) space -- measured slower than uniform splits; a no-op in the product build
    int merge_lists = 1;       // several lists per row are merged by their filter keys before the float64 re-evaluation
    int one_round = 1;         // fewer column-workgroups than slots: splits (+ a tail launch) chosen to fill ONE round of workgroups
    int l2c_geo = -1;          // pin the one-product kernel's geometry: 0 = A, 1 = B, 2 = C (-1: by list length)
    int l2_r1 = -1;            // fp32 filter: one row-block per wave (-1 auto, 0 never, 1 always)
    int split_refs_max = 0;    // lower the 2^25-references-per-split bound (tests see the rule at ordinary sizes)
    int cosine_centre = 1;     // cosine: centre the unit rows before packing (takes effect at the next set_ref)
    int coarse_kernel_q = 0;   // experiments: the one-product operands through the l2q kernel
    int order_flags = 0;       // experiments: locality-ordered streaming (order.hip)
};

struct OptionName { const char *name; int Options::*field; };
const OptionName OPTION_NAMES[] = {
    {"splits", &Options::splits}, {"tail_split", &Options::tail_split}, {"lkeep", &Options::lkeep},
    {"coarse_slack", &Options::coarse_slack}, {"cand_slack", &Options::cand_slack}, {"seeded_pass", &Options::seeded_pass},
    {"coarse_adapt", &Options::coarse_adapt}, {"wide_retry", &Options::wide_retry}, {"refine_overlap", &Options::refine_overlap},
    {"prepass", &Options::prepass}, {"pieces", &Options::pieces}, {"merge_lists", &Options::merge_lists}, {"one_round", &Options::one_round}, {"l2c_geo", &Options::l2c_geo}, {"l2_r1", &Options::l2_r1},
    {"split_refs_max", &Options::split_refs_max}, {"cosine_centre", &Options::cosine_centre},
    {"coarse_kernel_q", &Options::coarse_kernel_q}, {"order_flags", &Options::order_flags},
};

bool option_set(Options &o, const char *name, int64_t value)
{
    for (const OptionName &e : OPTION_NAMES)
        if (strcmp(e.name, name) == 0) {
            o.*(e.field) = (int)value;
            return true;
        }
    return false;
}

#ifdef NABO_EXPERIMENTS
void options_from_env(Options &o)
{
    for (const OptionName &e : OPTION_NAMES) {
        char key[64] = "NABO_OPT_";
        size_t k = strlen(key);
        for (const char *c = e.name; *c && k + 1 < sizeof(key); ++c) key[k++] = (char)(*c >= 'a' && *c <= 'z' ? *c - 32 : *c);
        key[k] = 0;
        const char *v = getenv(key);
        if (v && *v) o.*(e.field) = atoi(v);
    }
}
#endif

}  // namespace

struct nabo_index {
    int device = 0;
    int64_t n = 0;
    int g = 0;
    int metric = 0;
    double f = 0.25;
    int64_t base = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = {};
    // second stream: the refine of the main launch's rows runs beside the (short, split) tail launch of the filter
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_main = nullptr, ev_ref = nullptr;
    bool have_ref = false;

    const double *dY = nullptr;      // [n,g] float64 on device (borrowed or == ybuf)
    const double *dYp = nullptr;     // what the MFMA filter packs: dY, or the unit-length rows (cosine)
    DevBuf ybuf, ynbuf, xnbuf, maskbuf, mlistbuf;
    const uint8_t *dmask = nullptr;
    int64_t n_masked = 0;
    int n_masked_list = 0;
    bool shard_mode = false;       // set by nabo_sharded_query around its local queries: no masked tail (see tail_len)

    // Euclidean / cosine filter.  mode 0: fp32 MFMA only (l2_topk.hip); mode 1: f16x3 split on the f16 matrix pipe
    // (K-concatenated operands, kc steps of 16 slots; g < 64): l2h_topk.hip (per-wave streaming) or, when shared is set
    // and kc <= 10, l2s_topk.hip (reference tiles shared through an LDS ring) -- and the fp32 kernel for everything else
    int mode = 0;
    int kc = 0;
    bool shared = false;
    bool q16 = false;              // l2q_topk.hip (16x16x32 MFMA shape; operands in the layout16 packing)
    // One-product first pass (l2q kernel on [hi | norm | error] operands of kc1 steps, pack_ctiles_kernel<.,.,1>): rows it
    // cannot certify go through the f16x3 pass (no_coarse), then the 64-entry lists, then the exact kernels.
    // Passes of a query, each on the rows the one before could not certify (pass_level while query_impl recurses):
    //   0  one-product pass, lists built from +inf;  1  one-product pass SEEDED with the thresholds refine.hip derived
    //   from the failed certificates (l2c kernel only; 32-entry lists);  2  the f16x3 pass;  then the 64-entry lists
    //   (wide_retry) and the exact float64 kernels.
    bool coarse = false;
    // Set when the one-product passes of a query (>= 1024 rows) left more than a quarter of the rows to the pass behind
    // them: the bound 2^-9 ||x|| ||y|| is too weak for this reference set (tight clusters far from the centre of the data),
    // and the next queries start with the pass behind it right away.  Cleared by set_ref / set_mask.
    bool coarse_weak = false;
    int pass_level = 0;
    const float *seed_tau = nullptr;      // level 1: one threshold per row of the batch
    int kc1 = 0;
    double hscale = 1.0;
    double fscale = 1.0;           // power-of-two input scale of the fp32 path: max |y~| * fscale in (1/2, 1]
    int ksteps = 0;
    DevBuf centre, ypk, ycpk, ycpk1, normmax;
    // locality order (order.hip; l2q kernel only): reference keys / permutation (resident), target keys / permutation and
    // the waves' start tiles (per query), sort scratch
    bool order = false;            // decided at creation (NABO_L2Q_ORDER=0 streams in caller order: same results)
    int order_flags = 0;
    bool ref_ordered = false;      // the packed f16 tiles are in key order
    DevBuf rkeys, rperm, tkeys, tperm, wstart, okeys, opos, otemp;
    bool packed_f32 = false, packed_c16 = false, packed_c1 = false;
    int64_t ref_tiles = 0, ref_tiles_alloc = 0;
    double ymax_sqrt = 0.0, ymax_sqrt_c = 0.0;
    // Canberra path: exact kernel operands (yt) and the fp32 lower-bound filter's (ycf)
    DevBuf yt, ycf, yrow, cbflag, ych, cbscale, xh;    // ych/xh: 7-bit operands of the counting pass, cbscale [2g] doubles (min, 1/step)
    int cb_gp = 0;
    bool cb_f32 = false;          // filter usable for these references (fits fp32, g <= 128)
    // bit-sliced counting pass (canberra_bits.hip): quantile edges [g][B-1], cumulative bitmaps, valid bits, target row numbers
    DevBuf cbedges, cbtab, cbvalid, cbrow;
    bool cb_bits = false;
    int cb_mode = 0;              // NABO_CANBERRA_MODE at creation: 0 by size, 1 exact kernel only, 2 SWAR count, 3 bitmaps

    // query workspace
    DevBuf xfail, tmpi, tmpd, exact_d, fails2;
    DevBuf xfailp[2], tmpip[2], tmpdp[2], failsp[2], seedp[2], failseed;      // the same for passes 1 and 2 (the passes nest)
    DevBuf taupre, taupre2;                   // tournament seeds of the main / tail launch of the one-product pass [rows][S]
    DevBuf cand_key, cand_key2, cand_mi, cand_mt, cand_mi2, cand_mt2;   // filter keys of the lists; merged lists (merge_lists_kernel)
    DevBuf piecebuf;                          // a launch cut into pieces: pieces | ranges (int32)
    std::vector<int> piece_host;              // ... its host image (alive until the query's last synchronisation)
    int64_t pre_tiles_last = 0;               // reference tiles per split the last query's tournament looked at (0: none)
    int cand_slack = 3;                       // candidate mode on the one-product pass: kept entries beyond the emitted ones
    int64_t pass_rows[3] = {0, 0, 0};         // rows of the last query sent to the seeded pass / the f16x3 pass / the 64-entry lists
    // Which pass ANSWERED each row of the last top-level query (nabo_index_last_row_pass; NABO_PASS_* of nabo_knn.h): the
    // first filter's code for every row, overwritten as fail lists go down the chain.  row_map: rows of the batch a nested
    // query_impl works on -> rows of the top-level query (null at the top); depth: nesting level of query_impl.
    std::vector<uint8_t> row_pass;
    const std::vector<uint32_t> *row_map = nullptr;
    int depth = 0;
    float ms_keep[3] = {0, 0, 0};
    double ms_inner = 0.0;         // total of the most recent query_impl (read by the outer call of a retry)
    bool ms_keep_valid = false;
    bool wide_retry = false;       // inside the second-chance pass (64-entry lists for the rows the first pass could not certify)
    DevBuf xbuf, xpk, xnorm, cand_idx, cand_tau, cand_idx2, cand_tau2, cand_d, fails, failcnt, oidx, odist, nfound;
    int n_cu = 256;
    Options opt;

    double ms[5] = {0, 0, 0, 0, 0};
    int64_t counters[4] = {0, 0, 0, 0};
    char kernel[160] = "";          // dominant kernel of the last query (nabo_index_last_kernel)
    // nabo_index_query_async: the query runs on a host thread of its own (it synchronises its stream between its passes);
    // one in flight per index, joined by nabo_index_query_wait / any other call that needs the index
    std::thread async_thread;
    bool async_busy = false;
    int async_rc = NABO_OK;
    char async_msg[512] = "";
};

namespace nabo {
// for sharded.hip (same library, other translation unit)
int api_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int index_device(const nabo_index *ix) { return ix->device; }
int index_g(const nabo_index *ix) { return ix->g; }
int64_t index_n(const nabo_index *ix) { return ix->n; }
int index_metric(const nabo_index *ix) { return ix->metric; }
bool index_can_emit_candidates(const nabo_index *ix) { return ix->metric != NABO_METRIC_MOD_CANBERRA && ix->ksteps > 0; }
void index_set_shard_mode(nabo_index *ix, bool on) { ix->shard_mode = on; }
void index_set_cand_slack(nabo_index *ix, int s) { ix->cand_slack = s < 0 ? 0 : s; }
}  // namespace nabo


// Which filter kernels serve an index of this shape (n, g, metric are set): decided from the arguments and the mode string
// alone -- no device state -- so that nabo_query_plan can describe an index that does not exist.
//   mode (NABO_L2_MODE at nabo_index_create): unset / "f16x1" -- the DEFAULT chain: one-product pass (l2c_topk.hip, g <= 125)
//   -> seeded one-product pass -> f16x3 split (l2q_topk.hip, g < 64 and k' <= 28) or fp32-MFMA filter (l2_topk.hip) ->
//   64-entry lists -> exact float64 kernels;  "f16x3": the f16x3 split is the first pass;  "f32": the fp32-MFMA filter is.
//   (-DNABO_EXPERIMENTS builds also know "f16x3h", "f16x3s", "f16x1h": the 32x32x16 kernels of l2h_topk.hip / l2s_topk.hip.)
static void index_init_filters(nabo_index *ix, const char *md)
{
    const int g = ix->g;
    ix->mode = 0;
    ix->kc = ix->kc1 = 0;
    ix->shared = ix->coarse = ix->order = false;
    ix->q16 = true;
    ix->order_flags = 0;
    ix->ksteps = 0;
    if (ix->metric == NABO_METRIC_MOD_CANBERRA) return;
    ix->ksteps = pick_ksteps(g);           // -1: g > NABO_MAX_COMPS, every query takes the exact float64 route
    const bool f32 = md && strcmp(md, "f32") == 0, f16x3 = md && strncmp(md, "f16x3", 5) == 0;
    if (ix->ksteps <= 0) return;
    if (!f32 && nabo::l2q_pick_kc(g) > 0) {
        ix->mode = 1;                       // an f16x3 kernel exists for this g (g < 64)
        ix->kc = nabo::l2q_pick_kc(g);
#ifdef NABO_EXPERIMENTS
        ix->shared = md && strcmp(md, "f16x3s") == 0 && nabo::l2s_pick_kc(g) == ix->kc;
        ix->q16 = !(md && (strcmp(md, "f16x3h") == 0 || strcmp(md, "f16x3s") == 0 || strcmp(md, "f16x1h") == 0));
        // locality order (order.hip; option "order_flags", bit flags: 1 references, 2 targets in key order, 4 home pre-pass):
        // cuts the list updates by 30 % and the kernel is 20 % SLOWER on it (profiles/r3_order_experiment.txt)
        ix->order_flags = ix->q16 ? ix->opt.order_flags : 0;
        ix->order = (ix->order_flags & 1) != 0;
#endif
        ix->kc1 = ix->q16 ? nabo::l2c_pick_kc(g) : nabo::l2q_pick_kc1(g);
        ix->coarse = !ix->shared && ix->kc1 > 0 && !f16x3;
    } else if (!f32 && !f16x3 && nabo::l2c_pick_kc(g) > 0) {
        // 64 <= g <= 125: no f16x3 kernel is instantiated, but the one-product operands (g + 3 slots: four steps of 32)
        // are -- the one-product pass runs first, the fp32-MFMA filter takes the rows it cannot certify
        ix->kc1 = nabo::l2c_pick_kc(g);
        ix->coarse = true;
    }
}

// Entries of the masked-reference list a row may continue with when it has fewer than k' unmasked references
// (numpy.ma's NaN fill sorts the ignored references last, by index: nabo/_mapping.py:135-146).  A SHARD must not do
// that: its masked references would enter the global merge as if they were neighbours (found by the randomised
// sweep: 40-reference shards, 60 % masked) -- there the tail is left absent (index -1), which the merge skips.
static int tail_len(const nabo_index *ix) { return ix->shard_mode ? 0 : ix->n_masked_list; }

// The rows of the current batch listed in `d_rows` (device, nf entries) go on to the pass `code`: note it per top-level row and
// return their top-level row numbers in `map` (the inner query_impl's row_map).  The stream is synchronised.
static int note_row_pass(nabo_index *ix, const uint32_t *d_rows, int64_t nf, uint8_t code, std::vector<uint32_t> &map)
{
    map.resize((size_t)nf);
    if (nf == 0) return NABO_OK;
    HIP_TRY(hipMemcpyAsync(map.data(), d_rows, (size_t)nf * sizeof(uint32_t), hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    for (int64_t i = 0; i < nf; ++i) {
        if (ix->row_map) map[(size_t)i] = map[(size_t)i] < ix->row_map->size() ? (*ix->row_map)[map[(size_t)i]] : 0xFFFFFFFFu;
        if (map[(size_t)i] < ix->row_pass.size()) ix->row_pass[map[(size_t)i]] = code;
    }
    return NABO_OK;
}

#ifdef NABO_EXPERIMENTS
// Locality order of `n` rows of V (order.hip): sorted keys and the permutation, on the index's stream.
static int order_rows(nabo_index *ix, const double *V, int64_t n, DevBuf &keys, DevBuf &perm)
{
    int rc;
    size_t tb = 0;
    HIP_TRY(nabo::loc_sort_temp_bytes(n, nabo::loc_key_bits(ix->g), &tb));
    if ((rc = ix->okeys.reserve((size_t)n * 4)) || (rc = ix->opos.reserve((size_t)n * 4)) || (rc = keys.reserve((size_t)n * 4)) ||
        (rc = perm.reserve((size_t)n * 4)) || (rc = ix->otemp.reserve(tb + 16)))
        return rc;
    HIP_TRY(nabo::loc_order_launch(V, n, ix->g, ix->centre.as<double>(), ix->okeys.as<uint32_t>(), ix->opos.as<uint32_t>(),
                                   keys.as<uint32_t>(), perm.as<uint32_t>(), ix->otemp.p, tb, ix->stream));
    return NABO_OK;
}
#endif

// Pack the resident references for the fp32-MFMA kernel (want = 0), the f16x3 kernels (1: K-concatenated f16 tiles) or
// the one-product pass of the l2q kernel (2).
static int ensure_packed(nabo_index *ix, int want)
{
    const bool want_h = want != 0;
    if (want == 2 ? ix->packed_c1 : want == 1 ? ix->packed_c16 : ix->packed_f32) return NABO_OK;
    hipStream_t st = ix->stream;
    int rc;
    // normmax: [0] = max ||y~||^2 (float bits, SCALED units), [2..3] = max |y~ component| (double bits)
    if ((rc = ix->normmax.reserve(4 * sizeof(unsigned int)))) return rc;
    HIP_TRY(hipMemsetAsync(ix->normmax.p, 0, 4 * sizeof(unsigned int), st));
    unsigned int bits[4] = {0, 0, 0, 0};
    // power-of-two input scale from the largest centred component: the filter then works in a fixed numeric range
    // whatever the unit of the data (1e-30 or 1e+19 per component would under- / overflow fp32 squares otherwise)
    HIP_TRY(nabo::maxabs_launch(ix->dYp, ix->n, ix->g, ix->centre.as<double>(),
                                reinterpret_cast<unsigned long long *>(ix->normmax.as<unsigned int>() + 2), st));
    HIP_TRY(hipMemcpyAsync(bits, ix->normmax.p, sizeof(bits), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    double amax;
    memcpy(&amax, &bits[2], sizeof(amax));
    int e2 = 0;                                              // 2^e2 * amax in (1/2, 1]
    if (amax > 0 && std::isfinite(amax)) e2 = -(int)std::ceil(std::log2(amax));
    if (e2 > 480) e2 = 480;                                  // scale^2 must stay finite in float64
    if (e2 < -480) e2 = -480;
    double scale;
    if (want == 2) {
        if ((rc = ix->ycpk1.reserve((size_t)ix->ref_tiles_alloc * ix->kc1 * 1024 + 128))) return rc;
        ix->hscale = scale = std::ldexp(1.0, e2 + 12);
        ix->ref_ordered = false;
#ifdef NABO_EXPERIMENTS
        if (ix->order) {             // (the same keys, hence the same permutation, as the f16x3 operands of the second pass)
            if ((rc = order_rows(ix, ix->dYp, ix->n, ix->rkeys, ix->rperm))) return rc;
            ix->ref_ordered = true;
        }
#endif
        HIP_TRY(nabo::pack_cref_launch(ix->dYp, ix->n, ix->g, ix->centre.as<double>(), ix->hscale, ix->kc1,
                                       ix->ref_tiles_alloc, ix->dmask, ix->ycpk1.as<unsigned char>(),
                                       ix->normmax.as<unsigned int>(), ix->q16, st,
                                       ix->ref_ordered ? ix->rperm.as<uint32_t>() : nullptr, 1));
    } else if (want_h) {
        if ((rc = ix->ycpk.reserve((size_t)ix->ref_tiles_alloc * ix->kc * 1024 + 128))) return rc;
        // |v| <= 2^12 after scaling (f16 overflows at 65504; targets carry a factor 2)
        ix->hscale = scale = std::ldexp(1.0, e2 + 12);
        ix->ref_ordered = false;
#ifdef NABO_EXPERIMENTS
        if (ix->order) {
            if ((rc = order_rows(ix, ix->dYp, ix->n, ix->rkeys, ix->rperm))) return rc;
            ix->ref_ordered = true;
        }
#endif
        HIP_TRY(nabo::pack_cref_launch(ix->dYp, ix->n, ix->g, ix->centre.as<double>(), ix->hscale, ix->kc,
                                       ix->ref_tiles_alloc, ix->dmask, ix->ycpk.as<unsigned char>(),
                                       ix->normmax.as<unsigned int>(), ix->q16, st,
                                       ix->ref_ordered ? ix->rperm.as<uint32_t>() : nullptr));
    } else {
        const int Q = (ix->ksteps + 3) / 4;
        const size_t tile_bytes = ((size_t)Q * 256 + 32) * sizeof(float);
        if ((rc = ix->ypk.reserve((size_t)ix->ref_tiles_alloc * tile_bytes))) return rc;
        ix->fscale = scale = std::ldexp(1.0, e2);
        HIP_TRY(nabo::pack_ref_launch(ix->dYp, ix->n, ix->g, ix->centre.as<double>(), ix->fscale, ix->ksteps,
                                      ix->ref_tiles_alloc, ix->dmask, ix->ypk.as<float>(), ix->normmax.as<unsigned int>(), st));
    }
    HIP_TRY(hipMemcpyAsync(bits, ix->normmax.p, sizeof(bits), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    float fmax;
    memcpy(&fmax, &bits[0], sizeof(fmax));
    const double v = std::sqrt((double)fmax) / scale * (1.0 + 1e-6);      // unscaled units
    if (want == 2) { ix->ymax_sqrt_c = v; ix->packed_c1 = true; }
    else if (want_h) { ix->ymax_sqrt_c = v; ix->packed_c16 = true; }
    else { ix->ymax_sqrt = v; ix->packed_f32 = true; }
    return NABO_OK;
}

// mask + ascending list of the first masked indices (order-row tail, nabo/_mapping.py:135-144)
static int apply_mask(nabo_index *ix, const uint8_t *ref_mask)
{
    hipStream_t st = ix->stream;
    int rc;
    ix->dmask = nullptr;
    ix->n_masked = 0;
    ix->n_masked_list = 0;
    if (ref_mask) {
        std::vector<uint32_t> lst;
        for (int64_t j = 0; j < ix->n; ++j)
            if (ref_mask[j]) {
                ++ix->n_masked;
                lst.push_back((uint32_t)j);          // all of them: the exact route serves any k (order-row tail)
            }
        if (ix->n_masked > 0) {
            if ((rc = ix->maskbuf.reserve((size_t)ix->n))) return rc;
            HIP_TRY(hipMemcpyAsync(ix->maskbuf.p, ref_mask, (size_t)ix->n, hipMemcpyHostToDevice, st));
            ix->dmask = ix->maskbuf.as<uint8_t>();
            if ((rc = ix->mlistbuf.reserve(lst.size() * sizeof(uint32_t)))) return rc;
            HIP_TRY(hipMemcpyAsync(ix->mlistbuf.p, lst.data(), lst.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));      // lst goes out of scope
            ix->n_masked_list = (int)lst.size();
        }
    }
    return NABO_OK;
}

extern "C" {

const char *nabo_version(void) { return "nabo_knn 0.1 (gfx950)"; }
const char *nabo_last_error(void) { return g_err; }

int nabo_device_count(void)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt < 0 ? 0 : cnt;
}

int nabo_index_create(nabo_index **out, int32_t device, int64_t n_ref, int32_t g, int32_t metric,
                      double dist_factor, int64_t ref_index_base)
{
    if (!out) return fail(NABO_E_INVALID, "out is NULL");
    *out = nullptr;
    if (n_ref < 1 || n_ref >= 0xFFFFFFF0ll) return fail(NABO_E_INVALID, "n_ref=%lld out of range", (long long)n_ref);
    if (g < 1) return fail(NABO_E_INVALID, "g=%d must be >= 1", g);
    if (metric != NABO_METRIC_EUCLIDEAN && metric != NABO_METRIC_MOD_CANBERRA && metric != NABO_METRIC_COSINE)
        return fail(NABO_E_INVALID, "unknown metric %d", metric);
    if (metric == NABO_METRIC_MOD_CANBERRA && !(dist_factor > 0))
        return fail(NABO_E_INVALID, "dist_factor must be > 0");          // nabo/_mapping.py:516-521
    if (ref_index_base < 0) return fail(NABO_E_INVALID, "ref_index_base must be >= 0");
    // the shard merge carries global indices as 32-bit payloads (0xFFFFFFFF = absent)
    if (ref_index_base + n_ref > 0xFFFFFFFEll)
        return fail(NABO_E_UNSUPPORTED, "ref_index_base + n_ref = %lld exceeds 2^32 - 2", (long long)(ref_index_base + n_ref));
    int rc = use_device(device);
    if (rc) return rc;
    nabo_index *ix = new (std::nothrow) nabo_index();
    if (!ix) return fail(NABO_E_NOMEM, "host allocation failed");
    ix->device = device;
    ix->n = n_ref;
    ix->g = g;
    ix->metric = metric;
    ix->f = dist_factor;
    ix->base = ref_index_base;
#ifdef NABO_EXPERIMENTS
    options_from_env(ix->opt);
#endif
    {   // NABO_CANBERRA_MODE = exact | swar | bits pins the modified-Canberra path (default: by the size of the reference set)
        const char *cmode = getenv("NABO_CANBERRA_MODE");
        ix->cb_mode = !cmode ? 0 : strcmp(cmode, "exact") == 0 ? 1 : strcmp(cmode, "swar") == 0 ? 2 : strcmp(cmode, "bits") == 0 ? 3 : 0;
    }
    index_init_filters(ix, getenv("NABO_L2_MODE"));
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
        ix->n_cu = cus;
    hipError_t e = hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking);
    for (int i = 0; i < 6 && e == hipSuccess; ++i) e = hipEventCreate(&ix->ev[i]);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ix->stream2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ix->ev_main, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ix->ev_ref, hipEventDisableTiming);
    if (e != hipSuccess) {
        nabo_index_destroy(ix);
        return fail(NABO_E_HIP, "stream/event creation failed: %s", hipGetErrorString(e));
    }
    *out = ix;
    return NABO_OK;
}

int nabo_index_set_option(nabo_index *ix, const char *name, int64_t value)
{
    if (!ix || !name) return fail(NABO_E_INVALID, "NULL argument");
    if (ix->async_busy) return fail(NABO_E_INVALID, "an asynchronous query is in flight on this index: nabo_index_query_wait first");
    if (!option_set(ix->opt, name, value)) return fail(NABO_E_INVALID, "unknown option '%s'", name);
    if (strcmp(name, "order_flags") == 0) {
#ifdef NABO_EXPERIMENTS
        ix->order_flags = ix->q16 ? (int)value : 0;
        ix->order = (ix->order_flags & 1) != 0;
        ix->packed_f32 = ix->packed_c16 = ix->packed_c1 = false;
#else
        ix->opt.order_flags = 0;
        return fail(NABO_E_UNSUPPORTED, "option '%s' exists in -DNABO_EXPERIMENTS builds only", name);
#endif
    }
    return NABO_OK;
}

int nabo_index_destroy(nabo_index *ix)
{
    if (!ix) return NABO_OK;
    if (ix->async_thread.joinable()) ix->async_thread.join();     // (an asynchronous query still in flight: its buffers are the index's)
    (void)hipSetDevice(ix->device);
    if (ix->stream) (void)hipStreamSynchronize(ix->stream);
    for (int i = 0; i < 6; ++i)
        if (ix->ev[i]) (void)hipEventDestroy(ix->ev[i]);
    if (ix->stream2) { (void)hipStreamSynchronize(ix->stream2); (void)hipStreamDestroy(ix->stream2); }
    if (ix->ev_main) (void)hipEventDestroy(ix->ev_main);
    if (ix->ev_ref) (void)hipEventDestroy(ix->ev_ref);
    if (ix->stream) (void)hipStreamDestroy(ix->stream);
    delete ix;                       // every DevBuf member frees its allocation (the device is current)
    return NABO_OK;
}

int nabo_index_set_ref(nabo_index *ix, const double *Y, int32_t y_on_device, const uint8_t *ref_mask)
{
    if (!ix || !Y) return fail(NABO_E_INVALID, "NULL argument");
    if (ix->async_busy) return fail(NABO_E_INVALID, "an asynchronous query is in flight on this index: nabo_index_query_wait first");
    int rc = use_device(ix->device);
    if (rc) return rc;
    hipStream_t st = ix->stream;
    const size_t ybytes = (size_t)ix->n * ix->g * sizeof(double);
    if (y_on_device) {
        ix->dY = Y;
    } else {
        if ((rc = ix->ybuf.reserve(ybytes))) return rc;
        HIP_TRY(hipMemcpyAsync(ix->ybuf.p, Y, ybytes, hipMemcpyHostToDevice, st));
        ix->dY = ix->ybuf.as<double>();
    }
    if ((rc = apply_mask(ix, ref_mask))) return rc;
    if (ix->metric != NABO_METRIC_MOD_CANBERRA && ix->ksteps < 0) {
        HIP_TRY(hipStreamSynchronize(st));             // exact route only: the float64 rows are all it needs
    } else if (ix->metric != NABO_METRIC_MOD_CANBERRA) {
        ix->ref_tiles = (ix->n + 31) / 32;
        ix->ref_tiles_alloc = ix->ref_tiles + 64;      // room for split padding (+inf-norm tiles; up to 32 splits)
        ix->packed_f32 = ix->packed_c16 = ix->packed_c1 = false;
        ix->coarse_weak = false;
        if ((rc = ix->centre.reserve((size_t)ix->g * sizeof(double)))) return rc;
        if (ix->metric == NABO_METRIC_COSINE) {
            // cosine: the filter sees the unit-length rows x^, y^ and works on ||x^ - y^||^2 = 2 (1 - cos).  That quantity is
            // translation invariant like any Euclidean distance, so the UNIT rows are centred (a shift BEFORE the
            // normalisation would change angles; after it, it only shortens the vectors the error bounds scale with:
            // unit rows of PCA-like data sit in a cap around their mean direction, ||x^ - c|| is a fraction of 1)
            if ((rc = ix->ynbuf.reserve(ybytes))) return rc;
            HIP_TRY(nabo::normalise_rows_launch(ix->dY, ix->n, ix->g, ix->ynbuf.as<double>(), st));
            if (ix->opt.cosine_centre != 0)
                HIP_TRY(nabo::centre_launch(ix->ynbuf.as<double>(), ix->n, ix->g, ix->centre.as<double>(), st));
            else
                HIP_TRY(hipMemsetAsync(ix->centre.p, 0, (size_t)ix->g * sizeof(double), st));
            ix->dYp = ix->ynbuf.as<double>();
        } else {
            HIP_TRY(nabo::centre_launch(ix->dY, ix->n, ix->g, ix->centre.as<double>(), st));
            ix->dYp = ix->dY;
        }
        if ((rc = ensure_packed(ix, ix->coarse ? 2 : ix->mode == 1 ? 1 : 0))) return rc;
    } else {
        const int64_t chunks = (ix->n + 63) / 64;
        if ((rc = ix->yt.reserve((size_t)chunks * 64 * ix->g * sizeof(double)))) return rc;
        HIP_TRY(nabo::transpose_ref_launch(ix->dY, ix->n, ix->g, ix->yt.as<double>(), st));
        ix->cb_gp = nabo::cbf_pick_gp(ix->g);
        ix->cb_f32 = false;
        ix->cb_bits = false;
        if (ix->cb_gp > 0 && ix->cb_mode != 1) {
            unsigned int flag = 0;
            if ((rc = ix->ycf.reserve((size_t)chunks * 64 * ix->cb_gp * sizeof(float)))) return rc;      // chunk-major (range check)
            if ((rc = ix->yrow.reserve((size_t)ix->n * ix->cb_gp * sizeof(float)))) return rc;           // row-major (bound pass)
            if ((rc = ix->cbflag.reserve(4 * sizeof(unsigned int)))) return rc;
            HIP_TRY(hipMemsetAsync(ix->cbflag.p, 0, 4 * sizeof(unsigned int), st));
            HIP_TRY(nabo::cbf_pack_refs_launch(ix->dY, ix->n, ix->g, ix->cb_gp, ix->ycf.as<float>(), ix->cbflag.as<unsigned int>(), st));
            HIP_TRY(hipMemcpyAsync(&flag, ix->cbflag.p, sizeof(flag), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            ix->cb_f32 = (flag == 0);
            if (ix->cb_f32) {
                // per-dimension quantisation of the counting pass (canberra_f32.hip): min_k and 127 / (max_k - min_k) over the
                // references, from fp32 bounds that enclose every float64 value
                const int G = ix->g;
                std::vector<unsigned int> cm((size_t)2 * G, 0u);
                std::vector<double> sc((size_t)2 * G, 0.0);
                if ((rc = ix->cbscale.reserve((size_t)2 * G * sizeof(double)))) return rc;
                HIP_TRY(hipMemsetAsync(ix->cbscale.p, 0xFF, (size_t)G * sizeof(unsigned int), st));
                HIP_TRY(hipMemsetAsync(ix->cbscale.as<unsigned int>() + G, 0, (size_t)G * sizeof(unsigned int), st));
                HIP_TRY(nabo::cbf_colminmax_launch(ix->dY, ix->n, G, ix->cbscale.as<unsigned int>(), st));
                HIP_TRY(hipMemcpyAsync(cm.data(), ix->cbscale.p, (size_t)2 * G * sizeof(unsigned int), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                for (int k = 0; k < G; ++k) {
                    auto unord = [](unsigned int u) {
                        const unsigned int b = u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu);
                        float f;
                        memcpy(&f, &b, sizeof(f));
                        return (double)f;
                    };
                    const double lo = unord(cm[(size_t)k]), hi = unord(cm[(size_t)G + k]);
                    sc[(size_t)k] = std::isfinite(lo) ? lo : 0.0;
                    sc[(size_t)G + k] = (std::isfinite(lo) && std::isfinite(hi) && hi > lo && std::isfinite(127.0 / (hi - lo)))
                                            ? 127.0 / (hi - lo) : 0.0;              // 0: constant column, never counted as out
                }
                HIP_TRY(hipMemcpyAsync(ix->cbscale.p, sc.data(), (size_t)2 * G * sizeof(double), hipMemcpyHostToDevice, st));
                if ((rc = ix->ych.reserve((size_t)chunks * 64 * ix->cb_gp * 2))) return rc;
                HIP_TRY(nabo::cbf_pack_refs_rows_launch(ix->dY, ix->n, ix->g, ix->cb_gp, ix->yrow.as<float>(), st));
                HIP_TRY(nabo::cbf_pack_refs8_launch(ix->dY, ix->n, ix->g, ix->cb_gp, ix->cbscale.as<double>(), ix->ych.p, st));
                HIP_TRY(hipStreamSynchronize(st));      // sc goes out of scope
                // Bit-sliced counting pass (canberra_bits.hip), the default for reference sets of >= 12 blocks (25k cells; round 3, 128 blocks:
                // measured 2.0x the SWAR pass at 1M x 1M, level with it at 100k x 100k where building its table costs
                // what it saves; NABO_CANBERRA_MODE=swar pins the 7-bit SWAR pass, =bits the bitmaps at any size): per-dimension
                // QUANTILE bucket edges from a strided sample of the references (any edges give correct results -- they
                // only decide how sharp the count is), cumulative bitmaps per block of 2048 references.
                ix->cb_bits = false;
                // (round 4: from 12 blocks on -- with the four-words-per-lane kernel, its lists seeded before the count starts and
                // the edge sorts on eight host threads (set_ref 3.9 -> 1.2 ms) the bitmaps win from ~25k references even with
                // their table built inside the step: 100k x 100k 13.2 against 25.5 ms, 30k x 30k 3.0 against 3.6; 10k x 10k
                // 2.3 against 1.4 -- the same query time, the table build on top)
                const bool want_bits = ix->cb_mode ? ix->cb_mode == 3 : ix->n >= 12 * 2048;
                if (want_bits && nabo::cbb_available(G, ix->cb_gp, 1)) {
                    const int B = nabo::cbb_buckets();
                    int64_t ns = ix->n < 2048 ? ix->n : 2048;          // (32 sample values per bucket; the sort is host time inside set_ref)
                    const int64_t stride = ix->n / ns;
                    std::vector<double> smp((size_t)ns * G), col((size_t)ns), edges((size_t)G * (B - 1));
                    {   // the sample rows: gathered on the device, ONE contiguous copy back (a strided 2-D copy of 2048 short rows
                        // to pageable memory took milliseconds)
                        std::vector<uint32_t> rows_h((size_t)ns);
                        for (int64_t i = 0; i < ns; ++i) rows_h[(size_t)i] = (uint32_t)(i * stride);
                        if ((rc = ix->fails2.reserve((size_t)ns * sizeof(uint32_t)))) return rc;
                        if ((rc = ix->xfail.reserve((size_t)ns * G * sizeof(double)))) return rc;
                        HIP_TRY(hipMemcpyAsync(ix->fails2.p, rows_h.data(), (size_t)ns * sizeof(uint32_t), hipMemcpyHostToDevice, st));
                        HIP_TRY(nabo::gather_rows_launch(ix->dY, ix->fails2.as<uint32_t>(), ns, G, ix->xfail.as<double>(), st));
                        HIP_TRY(hipMemcpyAsync(smp.data(), ix->xfail.p, (size_t)ns * G * sizeof(double), hipMemcpyDeviceToHost, st));
                        HIP_TRY(hipStreamSynchronize(st));
                    }
                    // (one sort per dimension: a few host threads -- 50 sorts of 2048 values were 3.9 ms of every set_ref)
                    auto edges_of = [&](int k0, int k1) {
                        std::vector<double> colk((size_t)ns);
                        for (int k = k0; k < k1; ++k) {
                            size_t nf = 0;
                            for (int64_t i = 0; i < ns; ++i) {
                                const double v = smp[(size_t)i * G + k];
                                if (std::isfinite(v)) colk[nf++] = v;
                            }
                            std::sort(colk.begin(), colk.begin() + nf);
                            for (int b = 1; b < B; ++b)
                                edges[(size_t)k * (B - 1) + (b - 1)] = nf ? colk[(size_t)((double)b * nf / B)] : 0.0;
                        }
                    };
                    {
                        const int nt = G >= 16 ? 8 : 1;
                        std::vector<std::thread> th;
                        bool threaded = nt > 1;
                        if (threaded) {
                            try {
                                for (int t = 0; t < nt; ++t) th.emplace_back(edges_of, G * t / nt, G * (t + 1) / nt);
                            } catch (...) {
                                threaded = false;
                            }
                            for (std::thread &t : th) t.join();
                        }
                        if (!threaded) edges_of(0, G);           // (also after a failed thread start: every dimension again)
                    }
                    (void)col;
                    if ((rc = ix->cbedges.reserve(edges.size() * sizeof(double)))) return rc;
                    if ((rc = ix->cbtab.reserve(nabo::cbb_table_bytes(ix->n, G)))) return rc;
                    if ((rc = ix->cbvalid.reserve(nabo::cbb_valid_bytes(ix->n)))) return rc;
                    HIP_TRY(hipMemcpyAsync(ix->cbedges.p, edges.data(), edges.size() * sizeof(double), hipMemcpyHostToDevice, st));
                    HIP_TRY(nabo::cbb_pack_table_launch(ix->dY, ix->n, G, ix->cbedges.as<double>(), ix->cbtab.as<uint32_t>(), st));
                    HIP_TRY(nabo::cbb_valid_launch(ix->dmask, ix->n, ix->cbvalid.as<uint32_t>(), st));
                    HIP_TRY(hipStreamSynchronize(st));      // edges goes out of scope
                    ix->cb_bits = true;
                }
            }
        }
        HIP_TRY(hipStreamSynchronize(st));
    }
    ix->have_ref = true;
    return NABO_OK;
}

int nabo_index_set_mask(nabo_index *ix, const uint8_t *ref_mask)
{
    if (!ix) return fail(NABO_E_INVALID, "NULL index");
    if (ix->async_busy) return fail(NABO_E_INVALID, "an asynchronous query is in flight on this index: nabo_index_query_wait first");
    if (!ix->have_ref) return fail(NABO_E_INVALID, "nabo_index_set_ref has not been called");
    int rc = use_device(ix->device);
    if (rc) return rc;
    if ((rc = apply_mask(ix, ref_mask))) return rc;
    if (ix->metric == NABO_METRIC_MOD_CANBERRA && ix->cb_bits) {
        HIP_TRY(nabo::cbb_valid_launch(ix->dmask, ix->n, ix->cbvalid.as<uint32_t>(), ix->stream));
        HIP_TRY(hipStreamSynchronize(ix->stream));
    }
    if (ix->metric != NABO_METRIC_MOD_CANBERRA && ix->ksteps > 0) {       // masked cells carry ||y||^2 = +inf in the packed tiles
        ix->packed_f32 = ix->packed_c16 = ix->packed_c1 = false;
        ix->coarse_weak = false;
        if ((rc = ensure_packed(ix, ix->coarse ? 2 : ix->mode == 1 ? 1 : 0))) return rc;
    }
    return NABO_OK;
}

// ---- the launch plan of an Euclidean / cosine filter pass -------------------------------------------------------------------
// Everything the launch logic decides -- which kernel, in which geometry, how long the lists, how the reference range and the
// target rows are cut -- from the index's SHAPE and options and the query's shape alone: no device call, no state change.
// query_body executes it; nabo_query_plan returns it for an index that need not exist (tests/test_host_logic.py checks the
// rules on the CPU box: >= 256 workgroups whenever m x n allows, list lengths per pass, split bounds).
struct L2Plan {
    int epl = 1, L = 32;                 // emitted candidate lists hold L = 32 epl entries
    bool use_h = false;                  // an f16 kernel runs (one-product or f16x3 operands); false: the fp32-MFMA filter
    bool use_c = false;                  // experiments: the shared-tile f16x3 kernel
    bool use_1 = false;                  // one-product operands
    bool on_l2c = false;                 // ... on the l2c kernel (geo: its geometry)
    bool r1 = false;                     // fp32 filter: one row-block per wave
    int geo = -1, kcq = 0, slack1 = 0, cslack = 0;
    int rows_per_wg = 256, wg_per_cu = 1, lkeep_max = 32, lkeep = 16, want = 16;
    int S = 1, S2 = 1;                   // reference splits of the main / tail launch
    bool forced = false;                 // the split count is the caller's (option "splits")
    bool one_round = false;              // fewer column-workgroups than slots: splits (+ a tail launch on long streams) fill one round
    bool pieces = false;                 // the launch is cut into pieces (cut_pieces): piece_wgs workgroups of ~piece_len tiles, S = lists per row
    int piece_wgs = 0;
    int64_t piece_len = 0;
    int64_t gx = 0, gx_main = 0, gx_tail = 0, rows_pad = 0, tps = 0, tps2 = 0;
    char kernel[160] = "";
};

// A launch with fewer column-workgroups (gx) than the chip has slots, cut into pieces: the linear space (column, reference
// tile) of gx x T tiles goes to n_wg <= slots workgroups in equal chunks of ~C tiles (at least min_len: every piece warms
// its lists up on its own), a chunk that crosses a column boundary is two pieces, and a boundary that would leave a sliver of
// a column (< tiny tiles) is moved onto the column boundary.  Out: pieces (column, slot, t0, t1) in workgroup order,
// wg_first [chunks + 1] (which pieces a chunk holds), count [gx] pieces per column, *n_wg_out pieces in all (one workgroup
// each, l2c_topk.hip); returns the largest count (= lists per row, S).
static int cut_pieces(int64_t gx, int64_t T, int64_t slots, int64_t min_len, int max_per_col, std::vector<int> *pieces,
                      std::vector<int> *wg_first, std::vector<int> *count, int *n_wg_out, int64_t *len_out)
{
    const int64_t total = gx * T;
    // (a column is cut at most floor(T / C) + 2 ways: chunks long enough that a row never has more than max_per_col lists)
    if (max_per_col >= 3 && min_len < (T + max_per_col - 3) / (max_per_col - 2)) min_len = (T + max_per_col - 3) / (max_per_col - 2);
    int64_t n_wg = total / (min_len > 0 ? min_len : 1);
    if (n_wg > slots) n_wg = slots;
    if (n_wg < gx) n_wg = gx < slots ? gx : slots;
    if (n_wg < 1) n_wg = 1;
    int64_t C = (total + n_wg - 1) / n_wg;
    C = (C + 3) & ~(int64_t)3;
    if (C > T && gx >= n_wg) C = T;                       // (one column per workgroup at most when there is nothing to balance)
    const int64_t tiny = std::max<int64_t>(8, std::min<int64_t>(T / 4, C / 6));
    std::vector<int64_t> cutpos;                          // chunk boundaries in linear tile space
    cutpos.push_back(0);
    for (int64_t b = C; b < total; b += C) {
        int64_t bb = b;
        const int64_t pos = bb % T;
        if (pos != 0 && pos < tiny) bb -= pos;
        else if (pos != 0 && T - pos < tiny) bb += T - pos;
        if (bb > cutpos.back() && bb < total) cutpos.push_back(bb);
    }
    cutpos.push_back(total);
    if (pieces) pieces->clear();
    if (wg_first) wg_first->clear();
    std::vector<int> cnt((size_t)gx, 0);
    int np = 0, smax = 0;
    for (size_t w = 0; w + 1 < cutpos.size(); ++w) {
        if (wg_first) wg_first->push_back(np);
        int64_t lin = cutpos[w];
        while (lin < cutpos[w + 1]) {
            const int64_t col = lin / T, t0 = lin % T;
            const int64_t t1 = std::min<int64_t>(T, t0 + (cutpos[w + 1] - lin));
            const int slot = cnt[(size_t)col]++;
            if (pieces) { pieces->push_back((int)col); pieces->push_back(slot); pieces->push_back((int)t0); pieces->push_back((int)t1); }
            ++np;
            if (slot + 1 > smax) smax = slot + 1;
            lin += t1 - t0;
        }
    }
    if (wg_first) wg_first->push_back(np);
    if (count) *count = cnt;
    if (n_wg_out) *n_wg_out = np;                          // one workgroup per piece (chunks + column crossings)
    if (len_out) *len_out = C;
    return smax;
}
// shortest chunk of a cut launch, in reference tiles (30k x 30k, d = 50: four pieces of 234 tiles per column beat six and eight)
static const int64_t PIECE_MIN_TILES = 192;

static int plan_l2(const nabo_index *ix, int64_t m, int k, int drop, bool cand_mode, L2Plan *P)
{
    const int kk = k + drop;
    const int epl = ((kk <= 24 && !ix->wide_retry) || cand_mode) ? 1 : 2;
    const int L = 32 * epl;
    int rows_per_wg = 256, wg_per_cu = 1, lkeep_max = L;
    bool use_h = false, use_c = false;                   // use_h: an f16x3 kernel runs; use_c: the shared-tile one
    if (ix->mode == 1 && epl == 1) {
#ifdef NABO_EXPERIMENTS
        if (ix->shared) {
            nabo::l2s_topk_geometry(ix->kc, &rows_per_wg, &wg_per_cu, &lkeep_max);
            use_c = (cand_mode ? kk : kk + 4) <= lkeep_max;
        }
        if (!use_c && !ix->q16) nabo::l2h_topk_geometry(ix->kc, &rows_per_wg, &wg_per_cu, &lkeep_max);
        else
#endif
        if (!use_c) nabo::l2q_topk_geometry(ix->kc, &rows_per_wg, &wg_per_cu, &lkeep_max);
        use_h = (cand_mode ? kk : kk + 4) <= lkeep_max;   // needs at least 4 entries of slack
    }
    // The one-product first pass (kc1-step operands; see nabo_index::coarse) -- on the l2c kernel, in the geometry that
    // serves the list length the pass wants (l2c_topk.hip: two waves per SIMD up to 23 kept entries, 32-entry lists,
    // 64-entry lists for k' > 24), unless the locality order, the 32x32x16 experiment or an A/B run sends the operands
    // through the l2q / l2h kernel (those serve 32-entry lists and g < 64 only).
    const bool pass1 = ix->coarse && !(ix->coarse_weak && ix->opt.coarse_adapt != 0) && ix->pass_level < 2 &&
                       !ix->wide_retry && (!cand_mode || kk + 3 <= 32);
    // (k' > 24, the 64-entry lists: six entries more -- there a row the first pass fails is expensive, the pass behind the
    // seeded one is the fp32 filter: cosine 1M x 1M, d = 100, k = 50: 689 -> 597 ms per step)
    const int slack1 = ix->opt.coarse_slack >= 0 ? ix->opt.coarse_slack : (epl == 2 ? 6 : 0);
    const int cslack = ix->opt.cand_slack >= 0 ? ix->opt.cand_slack : ix->cand_slack;
    int want = cand_mode ? (kk < 4 ? 4 : kk) + cslack : kk + 8 + slack1;
    if (ix->pass_level == 1) want = L;                   // seeded pass: room for everything below the seed
    if (epl == 1 && want > 32) want = 32;                // (the emitted lists hold 32 epl entries)
    if (epl == 2) want = want < 33 ? 33 : (want > 64 ? 64 : want);
    int geo = -1;
    if (pass1 && ix->q16 && ix->order_flags == 0 && ix->opt.coarse_kernel_q == 0 && kk + 4 <= L) {
        geo = nabo::l2c_geometry(ix->kc1, want, ix->opt.l2c_geo);
        if (epl == 1 && geo == 2) geo = 0;               // (NABO_L2C_GEO=c with 32-entry emitted lists: geometry A)
    }
    const bool on_l2c = geo >= 0;
    const bool use_1 = on_l2c || (pass1 && use_h && !use_c);
    if (on_l2c) {
        use_h = true;
        use_c = false;
        nabo::l2c_topk_geometry(ix->kc1, want, ix->opt.l2c_geo, &rows_per_wg, &wg_per_cu, &lkeep_max);
        if (geo == 0) { rows_per_wg = 4 * 128; lkeep_max = 32; }
    }
    const int kcq = use_1 ? ix->kc1 : ix->kc;
    // fewer rows than two-row-block workgroups fill the chip with: one row-block per wave, three waves per SIMD
    // (128-row workgroups balance the CUs and the third wave covers the list warm-up that dominates short streams)
    bool r1 = false;
    if (!use_h) {
        nabo::l2_topk_geometry(ix->ksteps, epl, &rows_per_wg, &wg_per_cu, &lkeep_max);
        const int r1_mode = ix->opt.l2_r1;             // -1 auto, 0 never, 1 always (experiments)
        // ... and also when the list warm-up is a large share of a workgroup's time (short reference streams,
        // e.g. one shard of eight): the same per-workgroup model as the split choice below, threshold measured
        // (the variant pays ~8 % more per reference tile, it wins from ~7.5 % warm-up share on)
        int lk_est = cand_mode ? (kk < 4 ? 4 : kk) : (kk + 8 < 16 ? 16 : kk + 8);
        if (lk_est > L) lk_est = L;
        const double stream_ms = (double)((ix->n + 31) / 32) * 3.36e-3 * (ix->ksteps / 25.0);
        const double lg_est = std::log((double)ix->n / lk_est > 2.0 ? (double)ix->n / lk_est : 2.0);
        const double warm_ms = 5.1 * (lk_est / 24.0) * (lg_est / 10.6);
        if (epl == 1 && ix->ksteps <= 25 && r1_mode != 0 &&
            (r1_mode == 1 || (m + rows_per_wg - 1) / rows_per_wg < (int64_t)ix->n_cu * wg_per_cu ||
             warm_ms > 0.075 * stream_ms)) {
            r1 = true;
            nabo::l2_topk_geometry(ix->ksteps, -1, &rows_per_wg, &wg_per_cu, &lkeep_max);
        }
    }
    const int64_t slots = (int64_t)ix->n_cu * wg_per_cu;          // workgroups resident at once
    const int64_t gx = (m + rows_per_wg - 1) / rows_per_wg;
    const int64_t rows_pad = gx * rows_per_wg;
    // kept-list length: k' + 8 slack (the certification needs a gap above the k'-th distance)
    int lkeep = kk + 8;
    if (lkeep < 16) lkeep = 16;
    if (ix->wide_retry) lkeep = lkeep_max;              // as many kept entries as the 64-entry lists allow
    // one-product pass: its scores sit up to 2^-9 ||x|| ||y|| below the real ones and the gap above the k'-th distance
    // has to cover that -- 1M x 1M x 50: k' + 8 entries leave ~1 % of the rows to the f16x3 pass (7 ms), k' + 13 a
    // third of that, but every five entries more cost 14 ms of list updates in the kernel: no extra slack by default
    if (use_1) lkeep = kk + 8 + slack1;
    if (use_1 && ix->pass_level == 1) lkeep = lkeep_max;       // seeded pass: room for everything below the seed
    if (on_l2c && lkeep > want) lkeep = want;
    if (cand_mode) lkeep = kk < 4 ? 4 : kk;
    // (candidate mode on the one-product pass: three kept entries more than are emitted, so that the bound is the exact
    // distance of the first candidate left out and not the one-product threshold, which sits 2^-9 ||x|| ||y|| lower)
    if (cand_mode && use_1) lkeep += cslack;
    if (lkeep > lkeep_max) lkeep = lkeep_max;
    // (experiments and tests: the first pass's list length; the passes behind it keep theirs)
    if (ix->pass_level == 0 && !ix->wide_retry) { const int lk = ix->opt.lkeep; if (lk >= kk && lk <= lkeep_max) lkeep = lk; }
    // Work decomposition.  Few target rows: split the reference range S ways (grid.y) so the
    // chip is full.  Many rows: the last, partially filled round of workgroups is launched with
    // its own split factor S2 so that it takes ~1/S2 of a round instead of a whole one.
    int64_t gx_main = gx, gx_tail = 0;
    int S2 = 1;
    int S = ix->opt.splits;
    const bool forced = S > 0;
    bool pieces = false;
    int piece_wgs = 0;
    int64_t piece_len = 0;
#ifdef NABO_EXPERIMENTS                                     // (measured slower than uniform splits: section 4.6 of DESIGN.md; tools/ab builds only)
    if (!forced && on_l2c && ix->opt.pieces != 0 && ix->pass_level == 0 && !ix->wide_retry && gx < slots &&
        ix->ref_tiles >= 64 && ix->ref_tiles * 32 < NABO_LIST_SPLIT_REFS &&
        (ix->opt.split_refs_max < 64 || ix->ref_tiles * 32 < ix->opt.split_refs_max)) {
        const int sp = cut_pieces(gx, ix->ref_tiles, slots, PIECE_MIN_TILES, 1024 / L, nullptr, nullptr, nullptr, &piece_wgs, &piece_len);
        if (sp * L <= 1024) {                             // (merge / refine handle up to 1024 candidates per row)
            pieces = true;
            S = sp;
        }
    }
#endif
    // Fewer column-workgroups than slots, one-product kernel, lists merged before the float64 step (so a row's list count
    // costs the refine nothing): ONE round of workgroups at full occupancy -- all the columns with floor(slots / gx) uniform
    // splits when that fills at least 80 % of the slots.  On LONG reference streams (>= 8192 tiles) also one split more on
    // the floor(slots / S) columns that fit, the columns left over as a tail launch with more splits (the main / tail pair
    // of the long queries): 120k x 1M: 256 x 2 + 57 x 8, 13.3 instead of 16.1 ms.  On short streams a tail costs more than
    // the idle slots (100k x 100k: 256 x 2 + 5 x 16 behind the main launch 2.44 ms, beside it on the second stream 2.55,
    // 261 x 1 2.42), and so does cutting the (column, tile) space into equal chunks ("pieces", off): workgroups of a uniform
    // split stream the same tiles at the same time and share them in L2, unaligned pieces do not (49k x 100k: kernel
    // 1.99 ms as 603 pieces, 1.10 ms as 128 x 4).
    bool one_round = false;
    if (!forced && !pieces && on_l2c && ix->opt.one_round != 0 && ix->opt.merge_lists != 0 && ix->pass_level == 0 && !ix->wide_retry &&
        gx < slots) {
        int64_t s_cap = ix->ref_tiles / 16 > 0 ? ix->ref_tiles / 16 : 1;     // >= 16 tiles per split
        if (s_cap > 1024 / L) s_cap = 1024 / L;
        int64_t s_exact = slots / gx;
        if (s_exact > s_cap) s_exact = s_cap;
        S = (int)s_exact;
        const double occ = (double)(gx * s_exact) / (double)slots;
        if (occ < 0.8 && s_exact + 1 <= s_cap && ix->opt.tail_split != 0 && ix->ref_tiles >= 8192) {
            const int64_t s_up = s_exact + 1, cols = slots / s_up, rest = gx - cols;
            if (cols >= 1 && rest >= 1 && rest * 4 <= gx) {           // (the tail is a quarter of the columns at most)
                int64_t s2 = slots / rest;
                if (s2 > s_cap) s2 = s_cap;
                if (s2 > 16) s2 = 16;
                if (s2 < s_up) s2 = s_up;
                S = (int)s_up;
                S2 = (int)s2;
                gx_main = cols;
                gx_tail = rest;
            }
        }
        one_round = occ >= 0.8 || gx_tail > 0;               // (otherwise the cost model below decides)
    }
    if (!forced && !pieces && !one_round) {
        S = 1;
        if (gx < slots) {
            // Fewer workgroups than the chip holds: pick the split count from a cost model.  A workgroup costs
            // (reference tiles it streams) x t_tile for the MFMA chains PLUS a per-row list warm-up that does
            // not shrink with the stream (~lkeep * ln(stream / lkeep) appends per row: 5.1 ms per workgroup at
            // lkeep = 24 over 1M references, measured); every split pays the warm-up again.
            int64_t s_hi = ix->ref_tiles / 16 > 0 ? ix->ref_tiles / 16 : 1;
            if (s_hi > 1024 / L) s_hi = 1024 / L;
            // ms per reference tile and workgroup, measured: 105 ms / 31250 tiles (fp32, 256 rows, 25 k-steps); 1.2 us f16x3
            const double t_tile = use_c ? 0.7e-3 * ix->kc / 10.0 : use_h ? 1.1e-3 * kcq / 10.0
                                                                    : 3.36e-3 * (rows_per_wg / 256.0) * (ix->ksteps / 25.0);
            double best = 1e30;
            for (int s2 = 1; s2 <= (int)s_hi; ++s2) {
                const double rounds = (double)((gx * s2 + slots - 1) / slots);
                const double stream = (double)ix->n / s2;
                double lg = std::log(stream / lkeep > 2.0 ? stream / lkeep : 2.0);
                const double warm = 5.1 * (lkeep / 24.0) * (lg / 10.6) * (rows_per_wg / 256.0);
                const double cost = rounds * ((double)ix->ref_tiles / s2 * t_tile + warm);
                if (cost < best * (1.0 - 1e-3)) { best = cost; S = s2; }
            }
        } else if (gx % slots != 0 && ix->opt.tail_split != 0 && ix->ref_tiles >= 256) {
            const int64_t tail = gx % slots;
            double best = 1.0;
            // (at most 8 splits: 11 would fill the chip exactly at 1M x 1M -- kernel 0.6 ms shorter, refine of the tail
            // rows' 11 lists 1.1 ms longer)
            for (int s2 = 2; s2 <= 8; ++s2) {
                const double t = (double)((tail * s2 + slots - 1) / slots) / s2;
                if (t < best - 1e-9) { best = t; S2 = s2; }
            }
            if (S2 > 1) { gx_tail = tail; gx_main = gx - tail; }
        }
    }
    if (S > 1024 / L) S = 1024 / L;                     // refine merges at most 1024 candidates per row (32 or 16 lists)
    if (S < 1) S = 1;
    if ((int64_t)S > ix->ref_tiles) S = (int)ix->ref_tiles;
    // A seeded pass keeps at most L entries per list: where the first pass already kept (nearly) as many -- k' >= 43 on the
    // 64-entry lists: cosine d = 100, k = 50 -- one list per row certifies nothing the first pass could not.  Four
    // reference splits give a row four lists: the references below its seed (a few more than 64) spread over them.
    if (use_1 && ix->pass_level == 1 && !forced && kk + 8 + slack1 + 8 > L) {
        if (S < 4) S = 4;
        if (gx_tail > 0 && S2 < 4) S2 = 4;
    }
    {   // a list entry holds 25 bits of offset into its split (topk_lists.h): very large sets take more splits
        // (NABO_SPLIT_REFS_MAX: tests lower the bound to see the rule at ordinary sizes)
        int64_t split_refs = ix->opt.split_refs_max;
        if (split_refs < 64 || split_refs > NABO_LIST_SPLIT_REFS) split_refs = NABO_LIST_SPLIT_REFS;
        const int64_t split_tiles = (split_refs - 1) / 32;
        const int64_t s_min = (ix->ref_tiles + split_tiles - 1) / split_tiles;
        if (s_min > 1024 / L) return fail(NABO_E_INVALID, "more than 2^25 x (1024 / list length) reference cells in one index");
        if (S < s_min) S = (int)s_min;
        if (gx_tail > 0 && S2 < s_min) S2 = (int)s_min;
    }
    const int64_t tps = pieces ? ix->ref_tiles : (ix->ref_tiles + S - 1) / S;      // (pieces: the longest a piece can be)
    const int64_t tps2 = (ix->ref_tiles + S2 - 1) / S2;
    if ((!pieces && tps * S > ix->ref_tiles_alloc) || tps2 * S2 > ix->ref_tiles_alloc)
        return fail(NABO_E_INVALID, "internal: split padding exceeds allocation");

    P->epl = epl; P->L = L;
    P->use_h = use_h; P->use_c = use_c; P->use_1 = use_1; P->on_l2c = on_l2c; P->r1 = r1;
    P->geo = geo; P->kcq = kcq; P->slack1 = slack1; P->cslack = cslack;
    P->rows_per_wg = rows_per_wg; P->wg_per_cu = wg_per_cu; P->lkeep_max = lkeep_max; P->lkeep = lkeep; P->want = want;
    P->S = S; P->S2 = S2; P->forced = forced;
    P->pieces = pieces; P->piece_wgs = piece_wgs; P->piece_len = piece_len; P->one_round = one_round;
    P->gx = gx; P->gx_main = gx_main; P->gx_tail = gx_tail; P->rows_pad = rows_pad; P->tps = tps; P->tps2 = tps2;
    {
        // (the locality-ordered stream and NABO_COARSE_KERNEL_Q run the one-product operands through the l2q kernel)
        if (use_1 && ix->q16 && !on_l2c)
            snprintf(P->kernel, sizeof(P->kernel), "l2q_topk_kernel<%d,1,33> (v_mfma_f32_16x16x32_f16, one-product f16 filter with the split error as an operand slot)", kcq);
        else if (use_1 && ix->q16) snprintf(P->kernel, sizeof(P->kernel), "l2c_topk_kernel<%d,%s> (v_mfma_f32_16x16x32_f16, one-product f16 filter with the split error as an operand slot)", kcq / 2, geo == 1 ? "1,23,6,32,4,2" : geo == 2 ? "2,65,4,64,4,1" : "1,33,8,64,4,1");
        else if (use_1) snprintf(P->kernel, sizeof(P->kernel), "l2h_topk_kernel<%d,4,1,33> (v_mfma_f32_32x32x16_f16, one-product f16 filter with the split error as an operand slot)", kcq);
        else if (use_c) snprintf(P->kernel, sizeof(P->kernel), "l2s_topk_kernel<%d> (v_mfma_f32_32x32x16_f16, K-concatenated f16x3 split, LDS tile ring)", ix->kc);
        else if (use_h && ix->q16) snprintf(P->kernel, sizeof(P->kernel), "l2q_topk_kernel<%d,1,33> (v_mfma_f32_16x16x32_f16, K-concatenated f16x3 split)", ix->kc);
        else if (use_h) snprintf(P->kernel, sizeof(P->kernel), "l2h_topk_kernel<%d,4,1,33> (v_mfma_f32_32x32x16_f16, K-concatenated f16x3 split)", ix->kc);
        else snprintf(P->kernel, sizeof(P->kernel), "l2_topk_kernel<%d,%d,%d,%d> (v_mfma_f32_32x32x2_f32)", ix->ksteps,
                      r1 ? 1 : (epl == 1 ? 2 : 1), epl, epl == 1 ? 33 : 65);
    }
    return NABO_OK;
}

// cand_mode: shard mode of nabo_index_query_candidates -- k is the number of candidates per row to emit,
// out_bound [m] receives the squared-distance bound of everything not emitted; no local certification.
static int query_body(nabo_index *ix, const double *X, int32_t x_on_device, int64_t m, int32_t k, int32_t drop_first,
                      int64_t *out_idx, double *out_dist, int32_t out_on_device, bool cand_mode, double *out_bound);

static int query_impl(nabo_index *ix, const double *X, int32_t x_on_device, int64_t m, int32_t k, int32_t drop_first,
                      int64_t *out_idx, double *out_dist, int32_t out_on_device, bool cand_mode, double *out_bound)
{
    if (!ix) return fail(NABO_E_INVALID, "NULL argument");
    if (ix->depth == 0) {                            // a top-level query: the per-row record starts over
        ix->row_pass.clear();
        ix->row_map = nullptr;
    }
    ++ix->depth;
    const int rc = query_body(ix, X, x_on_device, m, k, drop_first, out_idx, out_dist, out_on_device, cand_mode, out_bound);
    --ix->depth;
    return rc;
}

static int query_body(nabo_index *ix, const double *X, int32_t x_on_device, int64_t m, int32_t k, int32_t drop_first,
                      int64_t *out_idx, double *out_dist, int32_t out_on_device, bool cand_mode, double *out_bound)
{
    if (!ix || !X || !out_idx || !out_dist) return fail(NABO_E_INVALID, "NULL argument");
    if (!ix->have_ref) return fail(NABO_E_INVALID, "nabo_index_set_ref has not been called");
    if (m < 0) return fail(NABO_E_INVALID, "m=%lld must be >= 0", (long long)m);
    if (m == 0) return NABO_OK;                      // no target cells: nothing to do (reference loops are empty)
    const int drop = drop_first ? 1 : 0;
    const int kk = k + drop;
    if (k < 1) return fail(NABO_E_INVALID, "k=%d must be >= 1", k);
    if (kk > ix->n && !cand_mode)
        return fail(NABO_E_INVALID, "k + drop_first = %d exceeds the %lld references", kk, (long long)ix->n);
    if (cand_mode && (ix->metric == NABO_METRIC_MOD_CANBERRA || !out_bound || !out_on_device || k > 32))
        return fail(NABO_E_INVALID, "candidate mode: Euclidean or cosine metric, device outputs, <= 32 candidates");
    // Shapes outside the instantiated filter kernels (k' > NABO_MAX_K, g > NABO_MAX_COMPS) are answered by the exact
    // float64 kernels for every row: the reference accepts any k / use_comps (nabo/_mapping.py:495-524).
    const bool exact_route = kk > NABO_MAX_K || (ix->metric != NABO_METRIC_MOD_CANBERRA && ix->ksteps < 0);
    if (exact_route && cand_mode)
        return fail(NABO_E_UNSUPPORTED, "candidate mode needs g <= %d (got %d)", NABO_MAX_COMPS, ix->g);
    int rc = use_device(ix->device);
    if (rc) return rc;
    hipStream_t st = ix->stream;
    const int g = ix->g;
    const int64_t n_valid = ix->n - ix->n_masked;

    // operands / results on device
    const double *dX = X;
    if (!x_on_device) {
        const size_t xb = (size_t)m * g * sizeof(double);
        if ((rc = ix->xbuf.reserve(xb))) return rc;
        HIP_TRY(hipMemcpyAsync(ix->xbuf.p, X, xb, hipMemcpyHostToDevice, st));
        dX = ix->xbuf.as<double>();
    }
    int64_t *d_oidx = out_idx;
    double *d_odist = out_dist;
    const size_t ob = (size_t)m * k * 8;
    if (!out_on_device) {
        if ((rc = ix->oidx.reserve(ob))) return rc;
        if ((rc = ix->odist.reserve(ob))) return rc;
        d_oidx = ix->oidx.as<int64_t>();
        d_odist = ix->odist.as<double>();
    }
    const int epl = ((kk <= 24 && !ix->wide_retry) || cand_mode) ? 1 : 2;
    const int L = 32 * epl;
    unsigned int n_fail = 0;
    int S = 1;
    int64_t n_wg = 0;
    HIP_TRY(hipEventRecord(ix->ev[0], st));
    const bool top = ix->depth == 1 && !cand_mode;   // this call owns the per-row pass record
    std::vector<uint32_t> pass_map;                  // top-level rows of the batch an inner call works on

    if (exact_route) {
        if (top) ix->row_pass.assign((size_t)m, (uint8_t)NABO_PASS_EXACT);
        uint64_t d_rows = (1ull << 30) / ((uint64_t)ix->n * sizeof(double));
        if (d_rows < 1) d_rows = 1;
        if (d_rows > (uint64_t)m) d_rows = (uint64_t)m;
        if (d_rows > 65528) d_rows = 65528;
        if (m > 0xFFFFFFF0ll) return fail(NABO_E_UNSUPPORTED, "m=%lld: fewer than 2^32-16 rows per call", (long long)m);
        if ((rc = ix->fails.reserve((size_t)m * sizeof(uint32_t)))) return rc;
        if ((rc = ix->exact_d.reserve((size_t)d_rows * ix->n * sizeof(double)))) return rc;
        HIP_TRY(nabo::iota_launch(ix->fails.as<uint32_t>(), m, st));
        HIP_TRY(hipEventRecord(ix->ev[1], st));
        HIP_TRY(hipEventRecord(ix->ev[2], st));
        HIP_TRY(hipEventRecord(ix->ev[3], st));
        HIP_TRY(nabo::exact_rows_launch(dX, ix->dY, ix->n, g, ix->metric, ix->f, ix->dmask, ix->fails.as<uint32_t>(),
                                        (unsigned int)m, k, drop, ix->base, ix->mlistbuf.as<uint32_t>(), tail_len(ix),
                                        d_oidx, d_odist, ix->exact_d.as<double>(), (unsigned int)d_rows, st));
        HIP_TRY(hipEventRecord(ix->ev[4], st));
        n_fail = (unsigned int)m;
        S = 0;
        snprintf(ix->kernel, sizeof(ix->kernel), "exact_dist_rows_kernel + exact_select_rows_kernel (float64 brute force)");
    } else if (ix->metric != NABO_METRIC_MOD_CANBERRA) {
        const bool cosine = ix->metric == NABO_METRIC_COSINE;
        const double *dXp = dX;                           // what the filter packs
        if (cosine) {
            if ((rc = ix->xnbuf.reserve((size_t)m * g * sizeof(double)))) return rc;
            HIP_TRY(nabo::normalise_rows_launch(dX, m, g, ix->xnbuf.as<double>(), st));
            dXp = ix->xnbuf.as<double>();
        }
        L2Plan P;
        if ((rc = plan_l2(ix, m, k, drop, cand_mode, &P))) return rc;
        const bool use_h = P.use_h, use_c = P.use_c, use_1 = P.use_1, on_l2c = P.on_l2c, forced = P.forced;
        const int geo = P.geo, kcq = P.kcq, slack1 = P.slack1, lkeep = P.lkeep, rows_per_wg = P.rows_per_wg, S2 = P.S2;
        const int epl_launch = P.r1 ? -1 : epl;
        const int64_t gx_main = P.gx_main, gx_tail = P.gx_tail, rows_pad = P.rows_pad, tps = P.tps, tps2 = P.tps2;
        const int Q = (ix->ksteps + 3) / 4;
        S = P.S;
        (void)forced; (void)slack1;
        if (ix->pass_level == 0 && !ix->wide_retry) {
            ix->pass_rows[0] = ix->pass_rows[1] = ix->pass_rows[2] = 0;
            snprintf(ix->kernel, sizeof(ix->kernel), "%s", P.kernel);
        }
        if ((rc = ensure_packed(ix, use_1 ? 2 : use_h ? 1 : 0))) return rc;
        if (top) ix->row_pass.assign((size_t)m, (uint8_t)(use_1 ? NABO_PASS_ONE_PRODUCT : NABO_PASS_SECOND));
        const int64_t rows_main = gx_main * rows_per_wg, rows_tail = gx_tail * rows_per_wg;
        const size_t xtile_bytes = use_h ? (size_t)kcq * 1024 : (size_t)Q * 256 * sizeof(float);
        if ((rc = ix->xpk.reserve((size_t)(rows_pad / 32) * xtile_bytes))) return rc;
        if ((rc = ix->xnorm.reserve((size_t)m * sizeof(double)))) return rc;
        if ((rc = ix->cand_idx.reserve((size_t)rows_main * S * L * sizeof(uint32_t) + 16))) return rc;
        if ((rc = ix->cand_tau.reserve((size_t)rows_main * S * sizeof(float) + 16))) return rc;
        if (gx_tail > 0) {
            if ((rc = ix->cand_idx2.reserve((size_t)rows_tail * S2 * L * sizeof(uint32_t)))) return rc;
            if ((rc = ix->cand_tau2.reserve((size_t)rows_tail * S2 * sizeof(float)))) return rc;
        }
        if ((rc = ix->fails.reserve((size_t)m * sizeof(uint32_t)))) return rc;
        if ((rc = ix->failcnt.reserve(sizeof(unsigned int)))) return rc;
        HIP_TRY(hipMemsetAsync(ix->failcnt.p, 0, sizeof(unsigned int), st));
        // locality order (order.hip): the l2q kernel streams key-ordered references; the targets are packed in key order
        // too and every wave starts its stream at its rows' neighbourhood
        const bool ordered = use_h && !use_c && ix->q16 && ix->ref_ordered;
        const bool t_ordered = use_h && !use_c && ix->q16 && (ix->order_flags & 2) != 0;
        const uint32_t *rperm = nullptr, *tperm = nullptr;
        const int32_t *wstart = nullptr;
#ifdef NABO_EXPERIMENTS
        if (ordered) rperm = ix->rperm.as<uint32_t>();
        if (t_ordered) {
            if ((rc = order_rows(ix, dXp, m, ix->tkeys, ix->tperm))) return rc;
            tperm = ix->tperm.as<uint32_t>();
        }
        if (ordered && t_ordered && (ix->order_flags & 4) != 0) {
            const int64_t n_waves = rows_pad / 128;
            if ((rc = ix->wstart.reserve((size_t)n_waves * 4))) return rc;
            HIP_TRY(nabo::wave_start_launch(ix->tkeys.as<uint32_t>(), m, 128, ix->rkeys.as<uint32_t>(), ix->n, n_waves,
                                            ix->wstart.as<int32_t>(), st));
            wstart = ix->wstart.as<int32_t>();
        }
#else
        (void)ordered; (void)t_ordered;
#endif
        if (use_h)
            HIP_TRY(nabo::pack_cquery_launch(dXp, m, g, ix->centre.as<double>(), ix->hscale, kcq, rows_pad / 32,
                                             ix->xpk.as<unsigned char>(), ix->xnorm.as<double>(), ix->q16, st, tperm,
                                             use_1 ? 1 : 3));
        else
            HIP_TRY(nabo::pack_query_launch(dXp, m, g, ix->centre.as<double>(), ix->fscale, ix->ksteps, rows_pad / 32,
                                            ix->xpk.as<float>(), ix->xnorm.as<double>(), st));
        HIP_TRY(hipEventRecord(ix->ev[1], st));
        bool seedable = false;               // the l2c kernel ran: its failed rows can go through a seeded pass
        bool refine_beside_tail = false;
        bool merge_main = false, merge_tail = false;
        int merge_keep = 0, merge_L = 0, keep_tail = 0;
        bool merge_seeded = false;
        int64_t pieces_wgs = 0;
        float *key_main = nullptr, *key_tail = nullptr;
#ifdef NABO_EXPERIMENTS
        if (use_c) {
            if (gx_main > 0)
                HIP_TRY(nabo::l2s_topk_launch(ix->kc, ix->xpk.as<unsigned char>(), ix->ycpk.as<unsigned char>(), (int)tps, S,
                                              (int)gx_main, 0, lkeep, ix->cand_idx.as<uint32_t>(), nullptr,
                                              ix->cand_tau.as<float>(), st));
            if (gx_tail > 0)
                HIP_TRY(nabo::l2s_topk_launch(ix->kc, ix->xpk.as<unsigned char>(), ix->ycpk.as<unsigned char>(), (int)tps2, S2,
                                              (int)gx_tail, rows_main / 32, lkeep, ix->cand_idx2.as<uint32_t>(), nullptr,
                                              ix->cand_tau2.as<float>(), st));
        } else
#endif
        if (use_h && ix->q16) {
            const unsigned char *ytiles = use_1 ? ix->ycpk1.as<unsigned char>() : ix->ycpk.as<unsigned char>();
            // the l2q kernel on the one-product operands: A/B runs, and the locality-ordered stream (its home pre-pass)
            const bool coarse_on_q = !on_l2c;
            seedable = use_1 && !coarse_on_q && !cand_mode && ix->opt.seeded_pass != 0;
            const float *seeds = (seedable && ix->pass_level == 1) ? ix->seed_tau : nullptr;
            // Several lists per row (reference splits, pieces, the tail round): the l2c kernel also emits the entries' filter
            // keys and merge_lists_kernel reduces the lists to the ONE a single stream would have kept (refine.hip)
            // (first pass only: a seeded pass WANTS every list re-evaluated -- its rows have more than one list's worth of
            // references below their seeds: cosine d = 100, k = 50 with the merge there: 86 instead of 16 ms of later passes)
            // ... the SEEDED pass keeps up to 128: what lies below a seed is "a few more than one list", and
            // 128 candidates are two per lane for the float64 step where S x 32 were four to sixteen per lane, each walking
            // its own row (100k x 100k: refine of 108 rows' 1024 candidates 0.41 ms)
            const bool seeded_merge = ix->pass_level == 1 && !ix->wide_retry;          // (32- and 64-entry lists alike)
            const bool merging = on_l2c && ix->opt.merge_lists != 0 && !ix->wide_retry && (ix->pass_level == 0 || seeded_merge);
            merge_seeded = seeded_merge;
            merge_keep = seeded_merge ? (S * L < 128 ? S * L : 128) : lkeep;
            merge_L = seeded_merge ? merge_keep : L;
            merge_main = merging && S > 1 && gx_main > 0;
            merge_tail = merging && S2 > 1 && gx_tail > 0;
            if (merge_main) {
                if ((rc = ix->cand_key.reserve((size_t)rows_main * S * L * sizeof(float)))) return rc;
                if ((rc = ix->cand_mi.reserve((size_t)rows_main * merge_L * sizeof(uint32_t)))) return rc;
                if ((rc = ix->cand_mt.reserve((size_t)rows_main * sizeof(float)))) return rc;
                key_main = ix->cand_key.as<float>();
            }
            if (merge_tail) {
                if ((rc = ix->cand_key2.reserve((size_t)rows_tail * S2 * L * sizeof(float)))) return rc;
                if ((rc = ix->cand_mi2.reserve((size_t)rows_tail * (merge_seeded ? 128 : merge_L) * sizeof(uint32_t)))) return rc;
                if ((rc = ix->cand_mt2.reserve((size_t)rows_tail * sizeof(float)))) return rc;
                key_tail = ix->cand_key2.as<float>();
            }
            if (use_1 && !coarse_on_q && P.pieces) {
                // A launch cut into pieces (cut_pieces): the tables go to the device, unused list slots read as empty lists
                // with threshold +inf, every piece's tournament looks at its own first tiles.
                std::vector<int> pcs_h, first_h, count_h;
                int n_wg = 0;
                int64_t plen = 0;
                const int sp = cut_pieces(gx_main, ix->ref_tiles, (int64_t)ix->n_cu * P.wg_per_cu, PIECE_MIN_TILES, 1024 / L, &pcs_h, &first_h, &count_h, &n_wg, &plen);
                if (sp != S) return fail(NABO_E_INVALID, "internal: piece plan changed between planning and launch");
                std::vector<int> ranges_h((size_t)gx_main * S * 4, 0);
                const int pre_pct = ix->opt.prepass;
                bool any_pre = false;
                for (size_t i = 0; i + 3 < pcs_h.size(); i += 4) {
                    int pt = 0, gt = 2;
                    if (pre_pct > 0) nabo::l2c_pre_plan(kcq, lkeep, pcs_h[i + 3] - pcs_h[i + 2], pre_pct, &pt, &gt);
                    int *r = &ranges_h[((size_t)pcs_h[i] * S + pcs_h[i + 1]) * 4];
                    r[0] = pcs_h[i + 2]; r[1] = pcs_h[i + 3]; r[2] = pt; r[3] = gt;
                    any_pre = any_pre || pt > 0;
                }
                // one workgroup per piece, longest first (the slots that finish a short piece pick up the next one)
                std::vector<int> order(pcs_h.size() / 4);
                for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
                std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
                    return pcs_h[4 * a + 3] - pcs_h[4 * a + 2] > pcs_h[4 * b + 3] - pcs_h[4 * b + 2];
                });
                std::vector<int> &img = ix->piece_host;        // (alive until the query's last synchronisation)
                img.clear();
                for (int i : order) img.insert(img.end(), pcs_h.begin() + 4 * i, pcs_h.begin() + 4 * i + 4);
                const size_t off_ranges = img.size();
                img.insert(img.end(), ranges_h.begin(), ranges_h.end());
                if ((rc = ix->piecebuf.reserve(img.size() * sizeof(int)))) return rc;
                HIP_TRY(hipMemcpyAsync(ix->piecebuf.p, img.data(), img.size() * sizeof(int), hipMemcpyHostToDevice, st));
                L2cPieces pcs;
                pcs.pieces = ix->piecebuf.as<int>();
                pcs.n_pieces = (int)order.size();
                pieces_wgs = pcs.n_pieces;
                pcs.ranges = ix->piecebuf.as<int>() + off_ranges;
                pcs.rows_per_col = rows_per_wg;
                HIP_TRY(hipMemsetAsync(ix->cand_idx.p, 0xFF, (size_t)rows_main * S * L * sizeof(uint32_t), st));
                HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ix->cand_tau.p), 0x7F800000, (size_t)rows_main * S, st));
                const float *seeds_main = nullptr;
                int stride_main = 0;
                if (ix->pass_level == 0 && !ix->wide_retry) ix->pre_tiles_last = 0;
                if (any_pre) {
                    if ((rc = ix->taupre.reserve((size_t)rows_main * S * sizeof(float)))) return rc;
                    HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ix->taupre.p), 0x7F800000, (size_t)rows_main * S, st));
                    HIP_TRY(nabo::l2c_pre_launch(kcq, lkeep, ix->xpk.as<unsigned char>(), ytiles, (int)tps, S, rows_main, 0, 0, 2,
                                                 ix->ref_tiles_alloc - 1, st, m, ix->taupre.as<float>(), pcs.ranges, rows_per_wg));
                    seeds_main = ix->taupre.as<float>();
                    stride_main = S;
                    if (ix->pass_level == 0 && !ix->wide_retry) ix->pre_tiles_last = ranges_h[2];
                }
                HIP_TRY(nabo::l2c_topk_launch(kcq, geo, ix->xpk.as<unsigned char>(), ytiles, (int)tps, S, (int)gx_main, 0, lkeep,
                                              ix->cand_idx.as<uint32_t>(), key_main, ix->cand_tau.as<float>(),
                                              ix->ref_tiles_alloc - 1, st, m, seeds_main, stride_main, 0, &pcs));
            } else if (use_1 && !coarse_on_q) {
                // Tournament seeds (l2c_topk.hip: l2c_pre_kernel): every (row, split) list starts from an upper bound of its
                // lkeep-th smallest score among the split's first references instead of +inf -- not for a pass that has its
                // seeds already.  NABO_PREPASS: 0 off, otherwise percent of the planned length (A/B runs; same bits always).
                const int pre_pct = seeds ? 0 : ix->opt.prepass;
                const float *seeds_main = seeds, *seeds_tail = seeds;
                int stride_main = 0, stride_tail = 0;
                if (ix->pass_level == 0 && !ix->wide_retry) ix->pre_tiles_last = 0;
                if (pre_pct > 0) {
                    int pt = 0, gt = 2;
                    nabo::l2c_pre_plan(kcq, lkeep, (int)tps, pre_pct, &pt, &gt);
                    if (pt > 0 && gx_main > 0) {
                        if ((rc = ix->taupre.reserve((size_t)rows_main * S * sizeof(float)))) return rc;
                        HIP_TRY(nabo::l2c_pre_launch(kcq, lkeep, ix->xpk.as<unsigned char>(), ytiles, (int)tps, S, rows_main, 0, pt, gt,
                                                     ix->ref_tiles_alloc - 1, st, m, ix->taupre.as<float>()));
                        seeds_main = ix->taupre.as<float>();
                        stride_main = S;
                        if (ix->pass_level == 0 && !ix->wide_retry) ix->pre_tiles_last = pt;
                    }
                    nabo::l2c_pre_plan(kcq, lkeep, (int)tps2, pre_pct, &pt, &gt);
                    if (pt > 0 && gx_tail > 0) {
                        if ((rc = ix->taupre2.reserve((size_t)rows_tail * S2 * sizeof(float)))) return rc;
                        HIP_TRY(nabo::l2c_pre_launch(kcq, lkeep, ix->xpk.as<unsigned char>(), ytiles, (int)tps2, S2, rows_tail,
                                                     rows_main / 32, pt, gt, ix->ref_tiles_alloc - 1, st, m, ix->taupre2.as<float>()));
                        seeds_tail = ix->taupre2.as<float>();
                        stride_tail = S2;
                    }
                }
                if (gx_main > 0)
                    HIP_TRY(nabo::l2c_topk_launch(kcq, geo, ix->xpk.as<unsigned char>(), ytiles, (int)tps, S, (int)gx_main, 0, lkeep,
                                                  ix->cand_idx.as<uint32_t>(), key_main, ix->cand_tau.as<float>(),
                                                  ix->ref_tiles_alloc - 1, st, m, seeds_main, stride_main, 0));
                // the tail launch (a fraction of a round, reference splits) leaves most CUs idle: the refine of the main
                // launch's rows (an HBM gather) runs beside it on the second stream
                if (gx_main > 0 && gx_tail > 0 && !cand_mode && ix->opt.refine_overlap != 0) {
                    HIP_TRY(hipEventRecord(ix->ev_main, st));
                    refine_beside_tail = true;
                }
                if (gx_tail > 0)
                    HIP_TRY(nabo::l2c_topk_launch(kcq, geo, ix->xpk.as<unsigned char>(), ytiles, (int)tps2, S2, (int)gx_tail,
                                                  rows_main / 32, lkeep, ix->cand_idx2.as<uint32_t>(), key_tail,
                                                  ix->cand_tau2.as<float>(), ix->ref_tiles_alloc - 1, st, m, seeds_tail, stride_tail,
                                                  rows_main));
            } else {
            if (gx_main > 0)
                HIP_TRY(nabo::l2q_topk_launch(kcq, ix->xpk.as<unsigned char>(), ytiles,
                                              (int)tps, S, (int)gx_main, 0, lkeep, ix->cand_idx.as<uint32_t>(), nullptr,
                                              ix->cand_tau.as<float>(), ix->ref_tiles_alloc - 1, st, wstart));
            if (gx_tail > 0)
                HIP_TRY(nabo::l2q_topk_launch(kcq, ix->xpk.as<unsigned char>(), ytiles,
                                              (int)tps2, S2, (int)gx_tail, rows_main / 32, lkeep,
                                              ix->cand_idx2.as<uint32_t>(), nullptr, ix->cand_tau2.as<float>(),
                                              ix->ref_tiles_alloc - 1, st, wstart));
            }
#ifdef NABO_EXPERIMENTS
        } else if (use_h) {
            const unsigned char *ytiles = use_1 ? ix->ycpk1.as<unsigned char>() : ix->ycpk.as<unsigned char>();
            if (gx_main > 0)
                HIP_TRY(nabo::l2h_topk_launch(kcq, ix->xpk.as<unsigned char>(), ytiles,
                                              (int)tps, S, (int)gx_main, 0, lkeep, ix->cand_idx.as<uint32_t>(), nullptr,
                                              ix->cand_tau.as<float>(), ix->ref_tiles_alloc - 1, st));
            if (gx_tail > 0)
                HIP_TRY(nabo::l2h_topk_launch(kcq, ix->xpk.as<unsigned char>(), ytiles,
                                              (int)tps2, S2, (int)gx_tail, rows_main / 32, lkeep,
                                              ix->cand_idx2.as<uint32_t>(), nullptr, ix->cand_tau2.as<float>(),
                                              ix->ref_tiles_alloc - 1, st));
#endif
        } else {
            if (gx_main > 0)
                HIP_TRY(nabo::l2_topk_launch(ix->ksteps, epl_launch, ix->xpk.as<float>(), ix->ypk.as<float>(), (int)tps, S,
                                             (int)gx_main, 0, lkeep, ix->cand_idx.as<uint32_t>(), nullptr,
                                             ix->cand_tau.as<float>(), st));
            if (gx_tail > 0)
                HIP_TRY(nabo::l2_topk_launch(ix->ksteps, epl_launch, ix->xpk.as<float>(), ix->ypk.as<float>(), (int)tps2, S2,
                                             (int)gx_tail, rows_main / 32, lkeep, ix->cand_idx2.as<uint32_t>(), nullptr,
                                             ix->cand_tau2.as<float>(), st));
        }
        HIP_TRY(hipEventRecord(ix->ev[2], st));
        if (nabo::debug_ablate() != 0) {     // -DNABO_EXPERIMENTS builds only: kernel-timing runs, results are garbage
            HIP_TRY(hipEventRecord(ix->ev[3], st));
            HIP_TRY(hipEventRecord(ix->ev[4], st));
            HIP_TRY(hipEventRecord(ix->ev[5], st));
            HIP_TRY(hipStreamSynchronize(st));
            float tt = 0;
            HIP_TRY(hipEventElapsedTime(&tt, ix->ev[1], ix->ev[2]));
            ix->ms[1] = tt;
            return NABO_OK;
        }
        // rounding-error coefficient of the filter score, relative to (||x|| + max||y||)^2
        // (f16x3: one fp32 accumulation per product term, 16 per step, plus the dropped lo*lo term and the
        // representation error of the hi + lo split)
        // (one-product pass: the hi x lo, lo x hi and lo x lo terms are INSIDE its score -- the error slot of
        // pack_ctiles_kernel<.,.,1> -- so the same accumulation / representation coefficient applies to its kc1 steps)
        const double err_coef = use_h ? 1.05 * ((16.0 * kcq + 8.0) * std::ldexp(1.0, -24) + std::ldexp(1.0, -20) + std::ldexp(1.0, -21))
                                      : 1.05 * (2.0 * ix->ksteps + 4.0) * std::ldexp(1.0, -24);
        const double tau_scale = use_h ? 1.0 / (ix->hscale * ix->hscale) : 1.0 / (ix->fscale * ix->fscale);
        const double ymax_sqrt = use_h ? ix->ymax_sqrt_c : ix->ymax_sqrt;
        const int64_t m_main = rows_main < m ? rows_main : m;
        // what the float64 re-evaluation reads: the filter's lists, or ONE merged list per row (merge_lists_kernel)
        const uint32_t *ci_main = ix->cand_idx.as<uint32_t>(), *ci_tail = ix->cand_idx2.as<uint32_t>();
        const float *ct_main = ix->cand_tau.as<float>(), *ct_tail = ix->cand_tau2.as<float>();
        int S_main = S, S_tail = S2, L_main = L, L_tail = L;
        auto merge_lists = [&](bool tail, hipStream_t sm) -> hipError_t {
            if (tail) {
                const int keep_t = merge_seeded ? (S2 * L < 128 ? S2 * L : 128) : merge_keep, lout_t = merge_seeded ? keep_t : merge_L;
                hipError_t e = nabo::merge_lists_launch(ix->cand_idx2.as<uint32_t>(), key_tail, ix->cand_tau2.as<float>(), m - rows_main, S2, L,
                                                        keep_t, lout_t, ix->cand_mi2.as<uint32_t>(), ix->cand_mt2.as<float>(), sm);
                ci_tail = ix->cand_mi2.as<uint32_t>(); ct_tail = ix->cand_mt2.as<float>(); S_tail = 1; L_tail = lout_t; keep_tail = keep_t;
                return e;
            }
            hipError_t e = nabo::merge_lists_launch(ix->cand_idx.as<uint32_t>(), key_main, ix->cand_tau.as<float>(), m_main, S, L, merge_keep,
                                                    merge_L, ix->cand_mi.as<uint32_t>(), ix->cand_mt.as<float>(), sm);
            ci_main = ix->cand_mi.as<uint32_t>(); ct_main = ix->cand_mt.as<float>(); S_main = 1; L_main = merge_L;
            return e;
        };
        if (cand_mode) {
            if (merge_main) HIP_TRY(merge_lists(false, st));
            if (merge_tail) HIP_TRY(merge_lists(true, st));
            HIP_TRY(nabo::refine_cand_launch(dX, 0, m_main, ix->dY, g, ci_main, ct_main,
                                             S_main, L_main, ix->xnorm.as<double>(), err_coef, ymax_sqrt, tau_scale, k, ix->base,
                                             n_valid, d_oidx, d_odist, out_bound, st, cosine ? 2 : 0, lkeep, rperm, tperm));
            if (gx_tail > 0)
                HIP_TRY(nabo::refine_cand_launch(dX, rows_main, m, ix->dY, g, ci_tail,
                                                 ct_tail, S_tail, L_tail, ix->xnorm.as<double>(), err_coef,
                                                 ymax_sqrt, tau_scale, k, ix->base, n_valid, d_oidx, d_odist, out_bound, st,
                                                 cosine ? 2 : 0, lkeep, rperm, tperm));
            HIP_TRY(hipEventRecord(ix->ev[3], st));
            HIP_TRY(hipEventRecord(ix->ev[4], st));
            HIP_TRY(hipEventRecord(ix->ev[5], st));
            HIP_TRY(hipStreamSynchronize(st));
            float tt = 0;
            for (int i = 0; i < 4; ++i) {
                HIP_TRY(hipEventElapsedTime(&tt, ix->ev[i], ix->ev[i + 1]));
                ix->ms[i] = tt;
            }
            HIP_TRY(hipEventElapsedTime(&tt, ix->ev[0], ix->ev[5]));
            ix->ms[4] = tt;
            ix->counters[0] = 0;
            ix->counters[1] = S;
            ix->counters[2] = L;
            ix->counters[3] = gx_main * S + gx_tail * S2;
            return NABO_OK;
        }
        float *fail_seed = nullptr;          // seeds for a seeded pass of the rows that fail (pass 0 on the l2c kernel)
        if (seedable && ix->pass_level == 0) {
            if ((rc = ix->failseed.reserve((size_t)m * sizeof(float)))) return rc;
            fail_seed = ix->failseed.as<float>();
        }
        hipStream_t st_main = st;
        if (refine_beside_tail) {
            st_main = ix->stream2;
            HIP_TRY(hipStreamWaitEvent(st_main, ix->ev_main, 0));
        }
        if (merge_main) HIP_TRY(merge_lists(false, st_main));
        HIP_TRY(nabo::refine_launch(dX, 0, m_main, ix->dY, g, ci_main, ct_main, S_main, L_main,
                                    ix->xnorm.as<double>(), err_coef, ymax_sqrt, tau_scale, k, drop, ix->base, n_valid,
                                    ix->mlistbuf.as<uint32_t>(), tail_len(ix), d_oidx, d_odist,
                                    ix->fails.as<uint32_t>(), ix->failcnt.as<unsigned int>(), st_main, cosine ? 2 : 0, 0.0, 0.0f,
                                    merge_main ? merge_keep : lkeep, rperm, tperm, fail_seed));
        if (refine_beside_tail) HIP_TRY(hipEventRecord(ix->ev_ref, st_main));
        if (merge_tail) HIP_TRY(merge_lists(true, st));
        if (gx_tail > 0)
            HIP_TRY(nabo::refine_launch(dX, rows_main, m, ix->dY, g, ci_tail,
                                        ct_tail, S_tail, L_tail, ix->xnorm.as<double>(), err_coef,
                                        ymax_sqrt, tau_scale, k, drop, ix->base, n_valid, ix->mlistbuf.as<uint32_t>(),
                                        tail_len(ix), d_oidx, d_odist, ix->fails.as<uint32_t>(),
                                        ix->failcnt.as<unsigned int>(), st, cosine ? 2 : 0, 0.0, 0.0f, merge_tail ? keep_tail : lkeep, rperm, tperm,
                                        fail_seed));
        if (refine_beside_tail) HIP_TRY(hipStreamWaitEvent(st, ix->ev_ref, 0));
        HIP_TRY(hipEventRecord(ix->ev[3], st));
        HIP_TRY(hipMemcpyAsync(&n_fail, ix->failcnt.p, sizeof(n_fail), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        float ms_first[3] = {0, 0, 0};
        bool retried = false;
        if (use_1 && n_fail > 0) {
            // Rows this one-product pass could not certify go on as a dense batch through this very function at the next
            // level: from level 0 on the l2c kernel to the SEEDED one-product pass (level 1: every row starts from the
            // threshold refine.hip derived from its failed certificate), otherwise to the f16x3 pass (level 2), which
            // sends what IT cannot certify on to the 64-entry lists / the exact kernels.
            for (int i = 0; i < 3; ++i) HIP_TRY(hipEventElapsedTime(&ms_first[i], ix->ev[i], ix->ev[i + 1]));
            const int64_t nf = n_fail;
            const int here = ix->pass_level;
            const int next = (here == 0 && fail_seed) ? 1 : 2;
            const int b = here;                              // buffer set of this frame (levels 0 and 1 recurse from here)
            if ((rc = ix->failsp[b].reserve((size_t)nf * sizeof(uint32_t)))) return rc;
            if ((rc = ix->xfailp[b].reserve((size_t)nf * g * sizeof(double)))) return rc;
            if ((rc = ix->tmpip[b].reserve((size_t)nf * k * sizeof(int64_t)))) return rc;
            if ((rc = ix->tmpdp[b].reserve((size_t)nf * k * sizeof(double)))) return rc;
            HIP_TRY(hipMemcpyAsync(ix->failsp[b].p, ix->fails.p, (size_t)nf * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
            if (next == 1) {
                if ((rc = ix->seedp[b].reserve((size_t)nf * sizeof(float)))) return rc;
                HIP_TRY(hipMemcpyAsync(ix->seedp[b].p, fail_seed, (size_t)nf * sizeof(float), hipMemcpyDeviceToDevice, st));
            }
            HIP_TRY(nabo::gather_rows_launch(dX, ix->failsp[b].as<uint32_t>(), nf, g, ix->xfailp[b].as<double>(), st));
            if ((rc = note_row_pass(ix, ix->failsp[b].as<uint32_t>(), nf, (uint8_t)(next == 1 ? NABO_PASS_SEEDED : NABO_PASS_SECOND), pass_map)))
                return rc;                                   // (synchronises the stream)
            const float *seed_saved = ix->seed_tau;
            const std::vector<uint32_t> *map_saved = ix->row_map;
            ix->pass_level = next;
            ix->seed_tau = next == 1 ? ix->seedp[b].as<float>() : nullptr;
            ix->row_map = &pass_map;
            rc = query_impl(ix, ix->xfailp[b].as<double>(), 1, nf, k, drop_first, ix->tmpip[b].as<int64_t>(),
                            ix->tmpdp[b].as<double>(), 1, false, nullptr);
            ix->pass_level = here;
            ix->seed_tau = seed_saved;
            ix->row_map = map_saved;
            if (rc) return rc;
            ix->pass_rows[next - 1] = nf;
            if (here == 0 && m >= 1024 && ix->pass_rows[1] > m / 4) ix->coarse_weak = true;
            n_fail = (unsigned int)ix->counters[0];          // rows that still needed the exact kernels
            HIP_TRY(nabo::scatter_rows_launch(ix->tmpip[b].as<int64_t>(), ix->tmpdp[b].as<double>(), ix->failsp[b].as<uint32_t>(),
                                              nf, k, d_oidx, d_odist, st));
            HIP_TRY(hipEventRecord(ix->ev[3], st));          // (ev[0..5] were reused by the inner call)
            retried = true;
        } else if (n_fail >= 16 && epl == 1 && !ix->wide_retry && ix->opt.wide_retry != 0) {
            // Second chance: rows the 32-entry lists could not certify (ties / near-ties reaching past the kept
            // entries) go through the same filter once more with 64-entry lists before anything is brute-forced.
            // The flagged rows are gathered into a dense batch; this very function solves it (wide_retry) and
            // sends what is STILL uncertified to the exact kernels; the answers are scattered back.
            for (int i = 0; i < 3; ++i) HIP_TRY(hipEventElapsedTime(&ms_first[i], ix->ev[i], ix->ev[i + 1]));
            const int64_t nf = n_fail;
            if ((rc = ix->fails2.reserve((size_t)nf * sizeof(uint32_t)))) return rc;
            if ((rc = ix->xfail.reserve((size_t)nf * g * sizeof(double)))) return rc;
            if ((rc = ix->tmpi.reserve((size_t)nf * k * sizeof(int64_t)))) return rc;
            if ((rc = ix->tmpd.reserve((size_t)nf * k * sizeof(double)))) return rc;
            HIP_TRY(hipMemcpyAsync(ix->fails2.p, ix->fails.p, (size_t)nf * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
            HIP_TRY(nabo::gather_rows_launch(dX, ix->fails2.as<uint32_t>(), nf, g, ix->xfail.as<double>(), st));
            if ((rc = note_row_pass(ix, ix->fails2.as<uint32_t>(), nf, (uint8_t)NABO_PASS_WIDE, pass_map))) return rc;      // (synchronises)
            const std::vector<uint32_t> *map_saved = ix->row_map;
            ix->wide_retry = true;
            ix->row_map = &pass_map;
            rc = query_impl(ix, ix->xfail.as<double>(), 1, nf, k, drop_first, ix->tmpi.as<int64_t>(), ix->tmpd.as<double>(), 1,
                            false, nullptr);
            ix->wide_retry = false;
            ix->row_map = map_saved;
            if (rc) return rc;
            ix->pass_rows[2] = nf;
            n_fail = (unsigned int)ix->counters[0];          // rows that still needed the exact kernels
            HIP_TRY(nabo::scatter_rows_launch(ix->tmpi.as<int64_t>(), ix->tmpd.as<double>(), ix->fails2.as<uint32_t>(), nf, k,
                                              d_oidx, d_odist, st));
            HIP_TRY(hipEventRecord(ix->ev[3], st));          // (ev[0..5] were reused by the inner call)
            retried = true;
        } else if (n_fail > 0) {
            // workspace for the exact distances of the flagged rows: up to ~1 GiB, at least one row
            uint64_t d_rows = (1ull << 30) / ((uint64_t)ix->n * sizeof(double));
            if (d_rows < 1) d_rows = 1;
            if (d_rows > n_fail) d_rows = n_fail;
            if (d_rows > 65535) d_rows = 65535;
            if ((rc = ix->exact_d.reserve((size_t)d_rows * ix->n * sizeof(double)))) return rc;
            if ((rc = note_row_pass(ix, ix->fails.as<uint32_t>(), n_fail, (uint8_t)NABO_PASS_EXACT, pass_map))) return rc;
            HIP_TRY(nabo::exact_rows_launch(dX, ix->dY, ix->n, g, ix->metric, ix->f, ix->dmask, ix->fails.as<uint32_t>(),
                                            n_fail, k, drop, ix->base, ix->mlistbuf.as<uint32_t>(), tail_len(ix),
                                            d_oidx, d_odist, ix->exact_d.as<double>(), (unsigned int)d_rows, st));
        }
        HIP_TRY(hipEventRecord(ix->ev[4], st));
        n_wg = pieces_wgs ? pieces_wgs : gx_main * S + gx_tail * S2;
        if (retried) { ix->ms_keep[0] = ms_first[0]; ix->ms_keep[1] = ms_first[1]; ix->ms_keep[2] = ms_first[2]; ix->ms_keep_valid = true; }
    } else {
        const int64_t n_chunks = (ix->n + 63) / 64;
        const int64_t gx = (m + 63) / 64;
        S = ix->opt.splits;
        if (S <= 0) {
            S = 1;
            if (gx < 512) {
                S = (int)((1024 + gx - 1) / gx);
                if (S > n_chunks) S = (int)n_chunks;
                if (S > 16) S = 16;
            }
        }
        if (S > 16) S = 16;
        if (S < 1) S = 1;
        bool done = false;
        const int S_exact = S;
        snprintf(ix->kernel, sizeof(ix->kernel), "canberra_topk_kernel (float64)");
        if (top) ix->row_pass.assign((size_t)m, (uint8_t)NABO_PASS_EXACT);
        if (ix->cb_f32 && n_valid >= kk) {
            // fp32 lower-bound filter -> float64 refine + certification -> exact re-solve of uncertified rows
            float slack, plateau;
            nabo::cbf_constants(g, &slack, &plateau);
            // which counting pass: bitmaps (canberra_bits.hip) where built and instantiated (32-entry lists), else SWAR
            const bool bits = ix->cb_bits && epl == 1;
            if (bits) snprintf(ix->kernel, sizeof(ix->kernel), "cbb_filter_kernel<%d> (bit-sliced count on %d-bucket bitmaps + fp32 lower bound)", ix->cb_gp, nabo::cbb_buckets());
            else snprintf(ix->kernel, sizeof(ix->kernel), "cbf_filter_kernel<%d> (7-bit integer count + fp32 lower bound)", ix->cb_gp);
            // filter geometry: T rows per workgroup, 2 workgroups per CU resident; every (row, split) ends
            // with `lists` candidate lists (one per wave).  Splits fill the chip when there are few rows and
            // trim the last, partially filled round of workgroups when there are many.
            const int lists = nabo::cbf_lists_per_split();
            const int rpw = bits ? nabo::cbb_rows_per_wg() : nabo::cbf_rows_per_wg(epl);
            const int64_t gxf = (m + rpw - 1) / rpw;
            // resident workgroups: SWAR pass -- one-wave workgroups, 2 per SIMD; bitmap pass -- ONE 8-wave workgroup per CU
            // (its LDS copy of the table rows + eight waves' lists fill the CU's LDS)
            const int64_t slots = bits ? (int64_t)ix->n_cu : (int64_t)ix->n_cu * 8;
            int Sf = ix->opt.splits;
            int s_max = 1024 / (lists * L);                   // refine handles <= 1024 candidates per row
            if (bits) {                                       // splits are ranges of 2048-reference blocks, >= 2 each
                const int64_t nb2 = ((ix->n + 2047) / 2048) / 2;
                if (s_max > nb2) s_max = (int)nb2;
            } else if (s_max > n_chunks / (8 * lists)) s_max = (int)(n_chunks / (8 * lists));
            if (s_max < 1) s_max = 1;
            if (Sf <= 0) {
                Sf = 1;
                double best = 1e30;
                for (int s2 = 1; s2 <= s_max; ++s2) {
                    // full-length rounds of workgroups, and ~8 % more bound evaluations per extra split
                    // (every list warms up on its own): measured on 100k x 100k, d = 50
                    const double cost = (double)((gxf * s2 + slots - 1) / slots) / s2 * (1.0 + 0.08 * (s2 - 1));
                    if (cost < best - 1e-9) { best = cost; Sf = s2; }
                }
            }
            if (Sf > s_max) Sf = s_max;
            // "tail round": with many rows the last, partially filled round of workgroups gets its own (larger)
            // split factor so that it takes a fraction of a round -- same idea as in the Euclidean launch above
            int64_t gx_main = gxf, gx_tail = 0;
            int S2 = 1;
            if (ix->opt.splits <= 0 && ix->opt.tail_split != 0 && gxf > slots && gxf % slots != 0 &&
                s_max >= 2) {
                const int64_t tail = gxf % slots;
                double best_t = 1e30;
                int best_s = 1;
                for (int s2 = 1; s2 <= s_max; ++s2) {
                    const double c = (double)((tail * s2 + slots - 1) / slots) / s2 * (1.0 + 0.08 * (s2 - 1));
                    if (c < best_t - 1e-9) { best_t = c; best_s = s2; }
                }
                const double cost_uniform = (double)((gxf * Sf + slots - 1) / slots) / Sf * (1.0 + 0.08 * (Sf - 1));
                const double cost_tail = (double)(gxf / slots) + best_t;
                if (best_s > 1 && cost_tail < cost_uniform - 1e-9) {
                    gx_tail = tail; gx_main = gxf - tail; S2 = best_s; Sf = 1;
                }
            }
            S = Sf;
            const int SL = Sf * lists, SL2 = S2 * lists;
            const int64_t rows_main = gx_tail > 0 ? gx_main * rpw : m;          // rows of the main launch
            const int64_t rows_tail = m - rows_main;
            if ((rc = ix->xpk.reserve((size_t)m * ix->cb_gp * 2 * sizeof(float)))) return rc;
            if ((rc = ix->xh.reserve((size_t)m * ix->cb_gp * 2))) return rc;
            if ((rc = ix->cand_idx.reserve((size_t)rows_main * SL * L * sizeof(uint32_t)))) return rc;
            if ((rc = ix->cand_tau.reserve((size_t)rows_main * SL * sizeof(float) + 16))) return rc;
            if (rows_tail > 0) {
                if ((rc = ix->cand_idx2.reserve((size_t)rows_tail * SL2 * L * sizeof(uint32_t)))) return rc;
                if ((rc = ix->cand_tau2.reserve((size_t)rows_tail * SL2 * sizeof(float) + 16))) return rc;
            }
            if ((rc = ix->fails.reserve((size_t)m * sizeof(uint32_t)))) return rc;
            const bool dbg_counts = (nabo::debug_ablate() & 4) != 0;
            if (dbg_counts) HIP_TRY(hipMemsetAsync(ix->cand_tau.as<float>() + (size_t)rows_main * SL, 0, 8, st));
            HIP_TRY(hipMemsetAsync(ix->cbflag.p, 0, 4 * sizeof(unsigned int), st));
            unsigned int *d_failcnt = ix->cbflag.as<unsigned int>() + 1, *d_flag = ix->cbflag.as<unsigned int>();
            HIP_TRY(nabo::cbf_pack_targets_launch(dX, m, g, ix->cb_gp, ix->f, ix->xpk.as<float>(), d_flag, st));
            if (bits) {
                if ((rc = ix->cbrow.reserve((size_t)m * ix->cb_gp * sizeof(uint16_t)))) return rc;
                HIP_TRY(nabo::cbb_pack_targets_launch(dX, m, g, ix->cb_gp, ix->f, ix->cbedges.as<double>(), ix->cbrow.as<uint16_t>(), st));
                HIP_TRY(hipEventRecord(ix->ev[1], st));
                HIP_TRY(nabo::cbb_filter_launch(ix->cb_gp, ix->xpk.as<float>(), ix->cbrow.as<uint16_t>(), rows_main,
                                                ix->yrow.as<float>(), ix->cbtab.as<uint32_t>(), ix->cbvalid.as<uint32_t>(), ix->n, g,
                                                Sf, ix->cand_idx.as<uint32_t>(), ix->cand_tau.as<float>(), st));
                if (rows_tail > 0)
                    HIP_TRY(nabo::cbb_filter_launch(ix->cb_gp, ix->xpk.as<float>() + (size_t)rows_main * ix->cb_gp * 2,
                                                    ix->cbrow.as<uint16_t>() + (size_t)rows_main * ix->cb_gp, rows_tail,
                                                    ix->yrow.as<float>(), ix->cbtab.as<uint32_t>(), ix->cbvalid.as<uint32_t>(),
                                                    ix->n, g, S2, ix->cand_idx2.as<uint32_t>(), ix->cand_tau2.as<float>(), st));
            } else {
                HIP_TRY(nabo::cbf_pack_targets8_launch(dX, m, g, ix->cb_gp, ix->f, ix->cbscale.as<double>(), ix->xh.p, st));
                HIP_TRY(hipEventRecord(ix->ev[1], st));
                HIP_TRY(nabo::cbf_filter_launch(ix->cb_gp, epl, ix->xpk.as<float>(), ix->xh.p, rows_main, ix->yrow.as<float>(),
                                                ix->ych.p, ix->n, g, ix->dmask, Sf, ix->cand_idx.as<uint32_t>(),
                                                ix->cand_tau.as<float>(), st));
                if (rows_tail > 0)
                    HIP_TRY(nabo::cbf_filter_launch(ix->cb_gp, epl, ix->xpk.as<float>() + (size_t)rows_main * ix->cb_gp * 2,
                                                    ix->xh.as<unsigned char>() + (size_t)rows_main * ix->cb_gp * 2, rows_tail,
                                                    ix->yrow.as<float>(), ix->ych.p, ix->n, g, ix->dmask, S2,
                                                    ix->cand_idx2.as<uint32_t>(), ix->cand_tau2.as<float>(), st));
            }
            HIP_TRY(hipEventRecord(ix->ev[2], st));
            if (dbg_counts) {
                unsigned int c2[2] = {0, 0};
                HIP_TRY(hipMemcpyAsync(c2, ix->cand_tau.as<float>() + (size_t)rows_main * SL, 8, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                fprintf(stderr, "[nabo debug] canberra filter (main launch): splits=%d survivors=%u (%.1f per row) batches=%u\n",
                        Sf, c2[0], (double)c2[0] / (double)rows_main, c2[1]);
            }
            HIP_TRY(nabo::refine_launch(dX, 0, rows_main, ix->dY, g, ix->cand_idx.as<uint32_t>(), ix->cand_tau.as<float>(), SL,
                                        L, nullptr, 0.0, 0.0, 1.0, k, drop, ix->base, n_valid, ix->mlistbuf.as<uint32_t>(),
                                        tail_len(ix), d_oidx, d_odist, ix->fails.as<uint32_t>(), d_failcnt, st, 1,
                                        ix->f, plateau));
            if (rows_tail > 0)
                HIP_TRY(nabo::refine_launch(dX, rows_main, m, ix->dY, g, ix->cand_idx2.as<uint32_t>(),
                                            ix->cand_tau2.as<float>(), SL2, L, nullptr, 0.0, 0.0, 1.0, k, drop, ix->base,
                                            n_valid, ix->mlistbuf.as<uint32_t>(), tail_len(ix), d_oidx, d_odist,
                                            ix->fails.as<uint32_t>(), d_failcnt, st, 1, ix->f, plateau));
            HIP_TRY(hipEventRecord(ix->ev[3], st));
            unsigned int hf[2] = {0, 0};
            HIP_TRY(hipMemcpyAsync(hf, ix->cbflag.p, sizeof(hf), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (hf[0] == 0) {                    // targets fit fp32: results stand, re-solve uncertified rows
                n_fail = hf[1];
                if (top) {
                    ix->row_pass.assign((size_t)m, (uint8_t)NABO_PASS_CANBERRA);
                    if ((rc = note_row_pass(ix, ix->fails.as<uint32_t>(), n_fail, (uint8_t)NABO_PASS_EXACT, pass_map))) return rc;
                }
                if (n_fail > 0) {
                    const int64_t nf = n_fail;
                    int S3 = 1;
                    const int64_t gx3 = (nf + 63) / 64;
                    if (gx3 < 512) {
                        S3 = (int)((1024 + gx3 - 1) / gx3);
                        if (S3 > n_chunks) S3 = (int)n_chunks;
                        if (S3 > 16) S3 = 16;
                    }
                    if ((rc = ix->xfail.reserve((size_t)nf * g * sizeof(double)))) return rc;
                    if ((rc = ix->cand_d.reserve((size_t)nf * S3 * L * sizeof(double)))) return rc;
                    if ((rc = ix->cand_idx2.reserve((size_t)nf * S3 * L * sizeof(uint32_t)))) return rc;
                    if ((rc = ix->tmpi.reserve((size_t)nf * k * sizeof(int64_t)))) return rc;
                    if ((rc = ix->tmpd.reserve((size_t)nf * k * sizeof(double)))) return rc;
                    HIP_TRY(nabo::gather_rows_launch(dX, ix->fails.as<uint32_t>(), nf, g, ix->xfail.as<double>(), st));
                    HIP_TRY(nabo::canberra_topk_launch(epl, ix->xfail.as<double>(), nf, ix->yt.as<double>(), ix->n, g, ix->f,
                                                       ix->dmask, S3, ix->cand_d.as<double>(), ix->cand_idx2.as<uint32_t>(), st));
                    HIP_TRY(nabo::merge_local_launch(ix->cand_d.as<double>(), ix->cand_idx2.as<uint32_t>(), nf, S3 * L, k, drop,
                                                     ix->base, ix->tmpi.as<int64_t>(), ix->tmpd.as<double>(), nullptr, st));
                    HIP_TRY(nabo::scatter_rows_launch(ix->tmpi.as<int64_t>(), ix->tmpd.as<double>(), ix->fails.as<uint32_t>(), nf,
                                                      k, d_oidx, d_odist, st));
                }
                HIP_TRY(hipEventRecord(ix->ev[4], st));
                done = true;
            }
        }
        if (!done) {
            S = S_exact;
            if ((rc = ix->cand_d.reserve((size_t)m * S * L * sizeof(double)))) return rc;
            if ((rc = ix->cand_idx.reserve((size_t)m * S * L * sizeof(uint32_t)))) return rc;
            HIP_TRY(hipEventRecord(ix->ev[1], st));
            HIP_TRY(nabo::canberra_topk_launch(epl, dX, m, ix->yt.as<double>(), ix->n, g, ix->f, ix->dmask, S,
                                               ix->cand_d.as<double>(), ix->cand_idx.as<uint32_t>(), st));
            HIP_TRY(hipEventRecord(ix->ev[2], st));
            HIP_TRY(nabo::merge_local_launch(ix->cand_d.as<double>(), ix->cand_idx.as<uint32_t>(), m, S * L, k, drop,
                                             ix->base, d_oidx, d_odist, nullptr, st));
            HIP_TRY(hipEventRecord(ix->ev[3], st));
            if (n_valid < kk)
                HIP_TRY(nabo::masked_tail_launch(dX, m, ix->dY, g, ix->metric, ix->f, ix->mlistbuf.as<uint32_t>(),
                                                 tail_len(ix), (int)n_valid, k, drop, ix->base, d_oidx, d_odist, st));
            HIP_TRY(hipEventRecord(ix->ev[4], st));
        }
        n_wg = gx * S;
    }
    if (!out_on_device) {
        HIP_TRY(hipMemcpyAsync(out_idx, d_oidx, ob, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(out_dist, d_odist, ob, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipEventRecord(ix->ev[5], st));
    HIP_TRY(hipStreamSynchronize(st));
    float t = 0;
    for (int i = 0; i < 4; ++i) {
        HIP_TRY(hipEventElapsedTime(&t, ix->ev[i], ix->ev[i + 1]));
        ix->ms[i] = t;
    }
    HIP_TRY(hipEventElapsedTime(&t, ix->ev[0], ix->ev[5]));
    ix->ms[4] = t;
    if (ix->ms_keep_valid) {            // a second-chance pass reused the events: phases 0-2 were saved before it ran
        ix->ms_keep_valid = false;
        HIP_TRY(hipEventElapsedTime(&t, ix->ev[3], ix->ev[5]));
        const double inner_total = ix->ms[4];      // (events 0..2 now belong to the inner call)
        (void)inner_total;
        ix->ms[0] = ix->ms_keep[0];
        ix->ms[1] = ix->ms_keep[1];
        ix->ms[2] = ix->ms_keep[2];
        ix->ms[3] = ix->ms_inner + t;              // inner pass (its own total) + scatter / copy-out
        ix->ms[4] = ix->ms[0] + ix->ms[1] + ix->ms[2] + ix->ms[3];
    }
    ix->ms_inner = ix->ms[4];
    ix->counters[0] = n_fail;
    ix->counters[1] = S;
    ix->counters[2] = L;
    ix->counters[3] = n_wg;
    return NABO_OK;
}

// (every entry point that touches an index first waits for the asynchronous query it may have in flight and hands its
// status to nabo_index_query_wait)
static void async_join(nabo_index *ix)
{
    if (ix && ix->async_thread.joinable()) ix->async_thread.join();
}

int nabo_index_query(nabo_index *ix, const double *X, int32_t x_on_device, int64_t m, int32_t k,
                     int32_t drop_first, int64_t *out_idx, double *out_dist, int32_t out_on_device)
{
    if (ix && ix->async_busy) return fail(NABO_E_INVALID, "an asynchronous query is in flight on this index: nabo_index_query_wait first");
    return query_impl(ix, X, x_on_device, m, k, drop_first, out_idx, out_dist, out_on_device, false, nullptr);
}

int nabo_index_query_async(nabo_index *ix, const double *X, int32_t x_on_device, int64_t m, int32_t k,
                           int32_t drop_first, int64_t *out_idx, double *out_dist, int32_t out_on_device)
{
    if (!ix) return fail(NABO_E_INVALID, "NULL argument");
    if (ix->async_busy) return fail(NABO_E_INVALID, "an asynchronous query is in flight on this index: nabo_index_query_wait first");
    async_join(ix);
    ix->async_busy = true;
    ix->async_rc = NABO_OK;
    ix->async_msg[0] = 0;
    try {
        ix->async_thread = std::thread([=]() {
            const int rc = query_impl(ix, X, x_on_device, m, k, drop_first, out_idx, out_dist, out_on_device, false, nullptr);
            ix->async_rc = rc;
            if (rc) snprintf(ix->async_msg, sizeof(ix->async_msg), "%s", nabo_last_error());   // (this thread's message)
        });
    } catch (...) {
        ix->async_busy = false;
        return fail(NABO_E_NOMEM, "could not start the query's host thread");
    }
    return NABO_OK;
}

int nabo_index_query_wait(nabo_index *ix)
{
    if (!ix) return fail(NABO_E_INVALID, "NULL argument");
    if (!ix->async_busy) return NABO_OK;
    async_join(ix);
    ix->async_busy = false;
    return ix->async_rc ? fail(ix->async_rc, "%s", ix->async_msg) : NABO_OK;
}

int nabo_index_query_candidates(nabo_index *ix, const double *X, int32_t x_on_device, int64_t m, int32_t n_cand,
                                int64_t *out_idx, double *out_dist, double *out_bound)
{
    if (ix && ix->async_busy) return fail(NABO_E_INVALID, "an asynchronous query is in flight on this index: nabo_index_query_wait first");
    return query_impl(ix, X, x_on_device, m, n_cand, 0, out_idx, out_dist, 1, true, out_bound);
}

int nabo_index_last_stats(const nabo_index *ix, double ms[5], int64_t counters[4])
{
    if (!ix) return fail(NABO_E_INVALID, "NULL index");
    if (ms) memcpy(ms, ix->ms, sizeof(ix->ms));
    if (counters) memcpy(counters, ix->counters, sizeof(ix->counters));
    return NABO_OK;
}

int nabo_index_last_passes(const nabo_index *ix, int64_t rows[3])
{
    if (!ix || !rows) return fail(NABO_E_INVALID, "NULL argument");
    rows[0] = ix->pass_rows[0];
    rows[1] = ix->pass_rows[1];
    rows[2] = ix->pass_rows[2];
    return NABO_OK;
}

int nabo_query_plan(int64_t n_ref, int32_t g, int32_t metric, int64_t m, int32_t k, int32_t drop_first, int32_t n_cand,
                    int32_t n_cu, const char *l2_mode, const char *options, int64_t out[NABO_PLAN_FIELDS], char *kernel,
                    size_t kernel_len)
{
    if (!out) return fail(NABO_E_INVALID, "NULL argument");
    if (n_ref < 1 || g < 1 || m < 1 || k < 1 || n_cu < 1) return fail(NABO_E_INVALID, "bad shape");
    if (metric != NABO_METRIC_EUCLIDEAN && metric != NABO_METRIC_COSINE)
        return fail(NABO_E_UNSUPPORTED, "nabo_query_plan describes the Euclidean / cosine filter launches");
    nabo_index ix;                                   // a shape, never a device object: nothing here touches HIP
    ix.n = n_ref;
    ix.g = g;
    ix.metric = metric;
    ix.n_cu = n_cu;
    if (options && *options) {                       // "name=value,name=value"
        char buf[512];
        snprintf(buf, sizeof(buf), "%s", options);
        for (char *tok = strtok(buf, ","); tok; tok = strtok(nullptr, ",")) {
            char *eq = strchr(tok, '=');
            if (!eq) return fail(NABO_E_INVALID, "option '%s': expected name=value", tok);
            *eq = 0;
            if (!option_set(ix.opt, tok, atoll(eq + 1))) return fail(NABO_E_INVALID, "unknown option '%s'", tok);
        }
    }
    index_init_filters(&ix, (l2_mode && *l2_mode) ? l2_mode : nullptr);
    ix.ref_tiles = (n_ref + 31) / 32;
    ix.ref_tiles_alloc = ix.ref_tiles + 64;
    const int drop = drop_first ? 1 : 0;
    const bool cand = n_cand > 0;
    const int kq = cand ? n_cand : k;
    for (int i = 0; i < NABO_PLAN_FIELDS; ++i) out[i] = 0;
    if (kq + (cand ? 0 : drop) > NABO_MAX_K || ix.ksteps < 0) {          // the exact float64 kernels answer every row
        out[0] = NABO_PASS_EXACT;
        out[1] = -1;
        if (kernel && kernel_len) snprintf(kernel, kernel_len, "exact_dist_rows_kernel + exact_select_rows_kernel (float64 brute force)");
        return NABO_OK;
    }
    L2Plan P;
    int rc = plan_l2(&ix, m, kq, cand ? 0 : drop, cand, &P);
    if (rc) return rc;
    int pt = 0, gt = 0;
    if (P.on_l2c && ix.opt.prepass > 0) nabo::l2c_pre_plan(P.kcq, P.lkeep, (int)(P.pieces ? P.piece_len : P.tps), ix.opt.prepass, &pt, &gt);
    out[0] = P.use_1 ? NABO_PASS_ONE_PRODUCT : NABO_PASS_SECOND;
    out[1] = P.geo;
    out[2] = P.rows_per_wg;
    out[3] = P.gx_main;
    out[4] = P.gx_tail;
    out[5] = P.S;
    out[6] = P.S2;
    out[7] = P.lkeep;
    out[8] = P.L;
    out[9] = P.tps;
    out[10] = pt;
    out[11] = gt;
    out[12] = (int64_t)n_cu * P.wg_per_cu;
    out[13] = P.pieces ? P.piece_wgs : P.gx_main * P.S + P.gx_tail * P.S2;
    out[14] = P.rows_pad;
    out[15] = P.kcq;
    out[16] = P.pieces ? 1 : 0;
    out[17] = P.piece_len;
    if (kernel && kernel_len) snprintf(kernel, kernel_len, "%s", P.kernel);
    return NABO_OK;
}

int nabo_index_last_row_pass(const nabo_index *ix, uint8_t *out, int64_t m)
{
    if (!ix || !out) return fail(NABO_E_INVALID, "NULL argument");
    if ((int64_t)ix->row_pass.size() != m)
        return fail(NABO_E_INVALID, "the last nabo_index_query on this index had %lld rows, not %lld (candidate queries keep no record)",
                    (long long)ix->row_pass.size(), (long long)m);
    if (m > 0) memcpy(out, ix->row_pass.data(), (size_t)m);
    return NABO_OK;
}

int nabo_index_last_kernel(const nabo_index *ix, char *buf, size_t n)
{
    if (!ix || !buf || n == 0) return fail(NABO_E_INVALID, "NULL argument");
    snprintf(buf, n, "%s", ix->kernel);
    return NABO_OK;
}

int nabo_knn(const double *X, int64_t m, const double *Y, int64_t n, int32_t g, int32_t k, int32_t metric,
             double dist_factor, const uint8_t *ref_mask, int32_t drop_first, int64_t *out_idx, double *out_dist,
             int32_t device)
{
    if (!X || !Y || !out_idx || !out_dist) return fail(NABO_E_INVALID, "NULL argument");
    nabo_index *ix = nullptr;
    int rc = nabo_index_create(&ix, device, n, g, metric, dist_factor, 0);
    if (rc) return rc;
    rc = nabo_index_set_ref(ix, Y, 0, ref_mask);
    if (!rc) rc = nabo_index_query(ix, X, 0, m, k, drop_first, out_idx, out_dist, 0);
    nabo_index_destroy(ix);
    return rc;
}

int nabo_pairwise(const double *X, int64_t m, const double *Y, int64_t n, int32_t g, int32_t metric,
                  double dist_factor, double *D, int32_t device)
{
    if (!X || !Y || !D) return fail(NABO_E_INVALID, "NULL argument");
    if (m < 1 || n < 1 || g < 1) return fail(NABO_E_INVALID, "empty operand");
    if (metric != NABO_METRIC_EUCLIDEAN && metric != NABO_METRIC_MOD_CANBERRA && metric != NABO_METRIC_COSINE)
        return fail(NABO_E_INVALID, "unknown metric %d", metric);
    if (m > 65535) return fail(NABO_E_UNSUPPORTED, "nabo_pairwise is the tile-sized seam: m <= 65535");
    int rc = use_device(device);
    if (rc) return rc;
    DevBuf dx, dy, dd;
    const size_t xb = (size_t)m * g * 8, yb = (size_t)n * g * 8, db = (size_t)m * n * 8;
    if ((rc = dx.reserve(xb)) || (rc = dy.reserve(yb)) || (rc = dd.reserve(db))) {
        dx.release(); dy.release(); dd.release();
        return rc;
    }
    hipError_t e = hipMemcpy(dx.p, X, xb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dy.p, Y, yb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = nabo::pairwise_launch(dx.as<double>(), m, dy.as<double>(), n, g, metric, dist_factor,
                                                   dd.as<double>(), nullptr);
    if (e == hipSuccess) e = hipMemcpy(D, dd.p, db, hipMemcpyDeviceToHost);
    dx.release(); dy.release(); dd.release();
    if (e != hipSuccess) return fail(NABO_E_HIP, "nabo_pairwise: %s", hipGetErrorString(e));
    return NABO_OK;
}

int nabo_merge_topk(int32_t device, const int64_t *parts_idx, const double *parts_dist, int32_t n_parts, int64_t m,
                    int32_t kp, int32_t k, int32_t drop_first, int64_t *out_idx, double *out_dist)
{
    if (!parts_idx || !parts_dist || !out_idx || !out_dist) return fail(NABO_E_INVALID, "NULL argument");
    const int drop = drop_first ? 1 : 0;
    if (n_parts < 1 || m < 1 || kp < 1 || k < 1) return fail(NABO_E_INVALID, "bad shape");
    if (k + drop > n_parts * kp) return fail(NABO_E_INVALID, "k + drop_first exceeds n_parts * kp");
    if ((int64_t)n_parts * kp > 1024) return fail(NABO_E_UNSUPPORTED, "n_parts * kp > 1024");
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(nabo::merge_parts_launch(parts_dist, parts_idx, n_parts, m, kp, k, drop, out_idx, out_dist, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return NABO_OK;
}

int nabo_snn_counts(int32_t device, const int64_t *t_idx, int64_t m, const int64_t *r_idx, int64_t n, int32_t k,
                    int32_t *out_snn)
{
    if (!t_idx || !r_idx || !out_snn) return fail(NABO_E_INVALID, "NULL argument");
    if (m < 1 || n < 1 || k < 1) return fail(NABO_E_INVALID, "bad shape");
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(nabo::snn_counts_launch(t_idx, m, r_idx, n, k, out_snn, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return NABO_OK;
}

// row_ptr != NULL: edges in CSR by reference node (edge_r unused); row_ptr == NULL: COO, E = n_edges, the CSR is built
// on the device by a stable sort (csr_build.hip).
static int score_null_impl(int32_t device, int64_t n_ref, const int64_t *row_ptr, int64_t n_edges, const int64_t *edge_r,
                           const int64_t *edge_t, const double *edge_w, int64_t n_t, const uint8_t *group, int32_t n_perm,
                           uint64_t seed, int32_t key_bits, double multiplier, double *out_obs, int64_t *out_nge,
                           double *out_mean, double *out_sd, int64_t *out_sizes)
{
    const bool coo = row_ptr == nullptr;
    if (!group || !out_obs || !out_nge || !out_mean || !out_sd) return fail(NABO_E_INVALID, "NULL argument");
    if (n_ref < 1 || n_t < 1) return fail(NABO_E_INVALID, "empty operand");
    if (n_perm < 1 || n_perm > 4096) return fail(NABO_E_UNSUPPORTED, "n_perm=%d: 1..4096 supported", n_perm);
    if (key_bits < 8 || key_bits > 64 || key_bits % 8) return fail(NABO_E_INVALID, "key_bits must be 8, 16, .., 64");
    if (n_ref >= 0x7FFFFFFFll) return fail(NABO_E_UNSUPPORTED, "n_ref too large for one launch");
    const int64_t E = coo ? n_edges : row_ptr[n_ref];
    if (coo) {
        if (E < 0) return fail(NABO_E_INVALID, "n_edges < 0");
        if (E >= 0xFFFFFFFFll) return fail(NABO_E_UNSUPPORTED, "n_edges=%lld: fewer than 2^32-1 supported", (long long)E);
        if (E > 0 && !edge_r) return fail(NABO_E_INVALID, "NULL edge arrays");
    } else {
        if (row_ptr[0] != 0 || E < 0) return fail(NABO_E_INVALID, "row_ptr must start at 0");
        for (int64_t r = 0; r < n_ref; ++r)
            if (row_ptr[r + 1] < row_ptr[r]) return fail(NABO_E_INVALID, "row_ptr must be non-decreasing");
        if (E > 0 && (!edge_t || !edge_w)) return fail(NABO_E_INVALID, "NULL edge arrays");
        for (int64_t e = 0; e < E; ++e)
            if (edge_t[e] < 0 || edge_t[e] >= n_t) return fail(NABO_E_INVALID, "edge_t[%lld] out of range", (long long)e);
    }
    if (E > 0 && (!edge_t || !edge_w)) return fail(NABO_E_INVALID, "NULL edge arrays");
    int64_t n_a = 0;
    for (int64_t t = 0; t < n_t; ++t) n_a += group[t] ? 1 : 0;
    if (n_a < 1) return fail(NABO_E_INVALID, "the group of interest is empty");
    int rc = use_device(device);
    if (rc) return rc;
    const int P = n_perm, W = (P + 1 + 31) / 32;
    hipStream_t st = nullptr;
    DevBuf d_rp, d_et, d_ew, d_grp, d_pre, d_hist, d_bits, d_nl, d_obs, d_nge, d_mean, d_sd;
    DevBuf c_r, c_t, c_w, c_ka, c_kb, c_pa, c_pb, c_tmp, c_flag;          // COO staging + sort scratch
    auto release = [&]() {
        DevBuf *all[] = {&d_rp, &d_et, &d_ew, &d_grp, &d_pre, &d_hist, &d_bits, &d_nl, &d_obs, &d_nge, &d_mean, &d_sd,
                         &c_r,  &c_t,  &c_w,  &c_ka,  &c_kb,  &c_pa,   &c_pb,   &c_tmp, &c_flag};
        for (DevBuf *b : all) b->release();
    };
#define NS_TRY(expr)                                                                     \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess) {                                                         \
            release();                                                                   \
            return fail(e__ == hipErrorOutOfMemory ? NABO_E_NOMEM : NABO_E_HIP, "%s failed: %s", #expr, \
                        hipGetErrorString(e__));                                         \
        }                                                                                \
    } while (0)
#define NS_RES(buf, bytes)                         \
    do {                                           \
        if ((rc = (buf).reserve(bytes))) {         \
            release();                             \
            return rc;                             \
        }                                          \
    } while (0)
    NS_RES(d_rp, (size_t)(n_ref + 1) * 8);
    NS_RES(d_et, (size_t)(E ? E : 1) * 8);
    NS_RES(d_ew, (size_t)(E ? E : 1) * 8);
    NS_RES(d_grp, (size_t)n_t);
    NS_RES(d_pre, (size_t)P * 8);
    NS_RES(d_hist, (size_t)P * 256 * 4);
    NS_RES(d_bits, (size_t)n_t * W * 4);
    NS_RES(d_nl, (size_t)P * 8);
    NS_RES(d_obs, (size_t)n_ref * 8);
    NS_RES(d_nge, (size_t)n_ref * 8);
    NS_RES(d_mean, (size_t)n_ref * 8);
    NS_RES(d_sd, (size_t)n_ref * 8);
    if (!coo) {
        NS_TRY(hipMemcpyAsync(d_rp.p, row_ptr, (size_t)(n_ref + 1) * 8, hipMemcpyHostToDevice, st));
        if (E) {
            NS_TRY(hipMemcpyAsync(d_et.p, edge_t, (size_t)E * 8, hipMemcpyHostToDevice, st));
            NS_TRY(hipMemcpyAsync(d_ew.p, edge_w, (size_t)E * 8, hipMemcpyHostToDevice, st));
        }
    } else {
        size_t tb = 0;
        NS_TRY(nabo::csr_sort_temp_bytes(E, n_ref, &tb));
        const size_t e1 = (size_t)(E ? E : 1);
        NS_RES(c_r, e1 * 8);
        NS_RES(c_t, e1 * 8);
        NS_RES(c_w, e1 * 8);
        NS_RES(c_ka, e1 * 4);
        NS_RES(c_kb, e1 * 4);
        NS_RES(c_pa, e1 * 4);
        NS_RES(c_pb, e1 * 4);
        NS_RES(c_tmp, tb ? tb : 1);
        NS_RES(c_flag, sizeof(unsigned int));
        if (E) {
            NS_TRY(hipMemcpyAsync(c_r.p, edge_r, (size_t)E * 8, hipMemcpyHostToDevice, st));
            NS_TRY(hipMemcpyAsync(c_t.p, edge_t, (size_t)E * 8, hipMemcpyHostToDevice, st));
            NS_TRY(hipMemcpyAsync(c_w.p, edge_w, (size_t)E * 8, hipMemcpyHostToDevice, st));
        }
        NS_TRY(nabo::csr_build_launch(c_r.as<int64_t>(), c_t.as<int64_t>(), c_w.as<double>(), E, n_ref, n_t,
                                      c_ka.as<uint32_t>(), c_pa.as<uint32_t>(), c_kb.as<uint32_t>(), c_pb.as<uint32_t>(),
                                      c_tmp.p, tb, d_rp.as<int64_t>(), d_et.as<int64_t>(), d_ew.as<double>(),
                                      c_flag.as<unsigned int>(), st));
        unsigned int bad = 0;
        NS_TRY(hipMemcpyAsync(&bad, c_flag.p, sizeof(bad), hipMemcpyDeviceToHost, st));
        NS_TRY(hipStreamSynchronize(st));
        if (bad) {
            release();
            return fail(NABO_E_INVALID, "%s out of range", (bad & 1u) ? "edge_ref" : "edge_t");
        }
        DevBuf *stage[] = {&c_r, &c_t, &c_w, &c_ka, &c_kb, &c_pa, &c_pb, &c_tmp};
        for (DevBuf *b : stage) b->release();
    }
    NS_TRY(hipMemcpyAsync(d_grp.p, group, (size_t)n_t, hipMemcpyHostToDevice, st));
    // radix select of the n_A-th smallest key of every permutation, 8 bits per pass
    std::vector<uint64_t> prefix((size_t)P, 0), below((size_t)P, 0), rank((size_t)P, (uint64_t)n_a);
    std::vector<int64_t> sizes((size_t)P, 0);
    std::vector<unsigned int> hist((size_t)P * 256);
    for (int done = 0; done < key_bits; done += 8) {
        NS_TRY(hipMemcpyAsync(d_pre.p, prefix.data(), (size_t)P * 8, hipMemcpyHostToDevice, st));
        NS_TRY(hipMemsetAsync(d_hist.p, 0, (size_t)P * 256 * 4, st));
        NS_TRY(nabo::null_hist_launch(n_t, P, seed, key_bits, d_pre.as<uint64_t>(), done, d_hist.as<unsigned int>(), st));
        NS_TRY(hipMemcpyAsync(hist.data(), d_hist.p, (size_t)P * 256 * 4, hipMemcpyDeviceToHost, st));
        NS_TRY(hipStreamSynchronize(st));
        for (int p = 0; p < P; ++p) {
            const unsigned int *h = &hist[(size_t)p * 256];
            uint64_t cum = 0;
            int b = 0;
            for (; b < 256; ++b) {
                if (cum + h[b] >= rank[p]) break;
                cum += h[b];
            }
            if (b == 256) { release(); return fail(NABO_E_HIP, "internal: radix select lost rank (permutation %d)", p); }
            prefix[p] = (prefix[p] << 8) | (uint64_t)b;
            below[p] += cum;
            rank[p] -= cum;
            if (done + 8 >= key_bits) sizes[p] = (int64_t)(below[p] + h[b]);      // keys <= T_p (ties at T_p included)
        }
    }
    NS_TRY(hipMemcpyAsync(d_pre.p, prefix.data(), (size_t)P * 8, hipMemcpyHostToDevice, st));        // thresholds T_p
    NS_TRY(hipMemcpyAsync(d_nl.p, sizes.data(), (size_t)P * 8, hipMemcpyHostToDevice, st));
    NS_TRY(nabo::null_label_launch(n_t, P, W, seed, key_bits, d_pre.as<uint64_t>(), d_grp.as<uint8_t>(), d_bits.as<uint32_t>(), st));
    NS_TRY(nabo::null_score_launch(n_ref, d_rp.as<int64_t>(), d_et.as<int64_t>(), d_ew.as<double>(), P, W,
                                   d_bits.as<uint32_t>(), d_nl.as<int64_t>(), n_a, multiplier, d_obs.as<double>(),
                                   d_nge.as<int64_t>(), d_mean.as<double>(), d_sd.as<double>(), st));
    NS_TRY(hipMemcpyAsync(out_obs, d_obs.p, (size_t)n_ref * 8, hipMemcpyDeviceToHost, st));
    NS_TRY(hipMemcpyAsync(out_nge, d_nge.p, (size_t)n_ref * 8, hipMemcpyDeviceToHost, st));
    NS_TRY(hipMemcpyAsync(out_mean, d_mean.p, (size_t)n_ref * 8, hipMemcpyDeviceToHost, st));
    NS_TRY(hipMemcpyAsync(out_sd, d_sd.p, (size_t)n_ref * 8, hipMemcpyDeviceToHost, st));
    NS_TRY(hipStreamSynchronize(st));
    if (out_sizes) memcpy(out_sizes, sizes.data(), (size_t)P * 8);
    release();
#undef NS_TRY
#undef NS_RES
    return NABO_OK;
}

int nabo_score_null(int32_t device, int64_t n_ref, const int64_t *row_ptr, const int64_t *edge_t,
                    const double *edge_w, int64_t n_t, const uint8_t *group, int32_t n_perm, uint64_t seed,
                    int32_t key_bits, double multiplier, double *out_obs, int64_t *out_nge, double *out_mean,
                    double *out_sd, int64_t *out_sizes)
{
    if (!row_ptr) return fail(NABO_E_INVALID, "NULL argument");
    return score_null_impl(device, n_ref, row_ptr, 0, nullptr, edge_t, edge_w, n_t, group, n_perm, seed, key_bits,
                           multiplier, out_obs, out_nge, out_mean, out_sd, out_sizes);
}

int nabo_score_null_edges(int32_t device, int64_t n_ref, int64_t n_edges, const int64_t *edge_ref, const int64_t *edge_t,
                          const double *edge_w, int64_t n_t, const uint8_t *group, int32_t n_perm, uint64_t seed,
                          int32_t key_bits, double multiplier, double *out_obs, int64_t *out_nge, double *out_mean,
                          double *out_sd, int64_t *out_sizes)
{
    return score_null_impl(device, n_ref, nullptr, n_edges, edge_ref, edge_t, edge_w, n_t, group, n_perm, seed, key_bits,
                           multiplier, out_obs, out_nge, out_mean, out_sd, out_sizes);
}

int nabo_dev_malloc(int32_t device, void **ptr, size_t bytes)
{
    if (!ptr) return fail(NABO_E_INVALID, "NULL argument");
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipMalloc(ptr, bytes ? bytes : 1));
    return NABO_OK;
}

int nabo_dev_free(int32_t device, void *ptr)
{
    int rc = use_device(device);
    if (rc) return rc;
    if (ptr) HIP_TRY(hipFree(ptr));
    return NABO_OK;
}

int nabo_memcpy_h2d(int32_t device, void *dst, const void *src, size_t bytes)
{
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return NABO_OK;
}

int nabo_memcpy_d2h(int32_t device, void *dst, const void *src, size_t bytes)
{
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return NABO_OK;
}

int nabo_dev_mem_info(int32_t device, size_t *free_bytes, size_t *total_bytes)
{
    if (!free_bytes || !total_bytes) return fail(NABO_E_INVALID, "NULL argument");
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipMemGetInfo(free_bytes, total_bytes));
    return NABO_OK;
}

int nabo_dev_synchronize(int32_t device)
{
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return NABO_OK;
}

}  // extern "C"
