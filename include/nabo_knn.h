/*
 * nabo_knn.h -- C ABI of libnabo_knn.so, the MI355X (gfx950) implementation of Nabo's
 * cross-sample k-NN mapping hot path.
 *
 * The reference (parashardhapola/nabo v0.4.1) has no FFI: the seam is a Python function
 * boundary inside nabo/_mapping.py.  Each entry point below names the reference code it
 * replaces (file:line relative to the reference root).  Plain pointers and sizes only;
 * no torch / numpy types.  All functions return 0 on success or a negative NABO_E_* code,
 * with a human-readable message available from nabo_last_error() (thread-local).
 *
 * Semantics shared by every k-NN entry point
 *   - distances are the reference's float64 values, bit for bit:
 *       metric 0: nabo/_mapping.py:16-26  _euclidean_dist   sqrt(sum_k (x-y)^2), k ascending,
 *                                         separate multiply/add (no FMA), correctly rounded sqrt
 *       metric 1: nabo/_mapping.py:29-45  _mod_canberra_dist (asymmetric: x = target, y = reference)
 *       metric 2: EXTENSION, not in the reference (BASELINE.json configs[4] names a cosine metric the
 *                 reference lacks; parity pinned only by this build's own oracle): cosine distance
 *                 1 - <x,y>/(sqrt<x,x>*sqrt<y,y>), sums in ascending k with separate multiply/add;
 *                 a zero vector is at distance 1 from everything
 *   - ordering replaces the mask + np.argsort of nabo/_mapping.py:135-146: refs flagged in
 *     ref_mask sort to the END (numpy.ma NaN-fill), all others by (distance ascending,
 *     reference index ascending) -- the canonical total order where the reference's unstable
 *     sort leaves ties undefined;
 *   - drop_first != 0 reproduces the positional `[1:]` of nabo/_mapping.py:142 (intra-reference);
 *   - out_idx / out_dist are [m, k] row-major: the first k entries of each order row (what
 *     _calc_snn reads, nabo/_mapping.py:190,193) and their distances.
 * There is NO CPU fallback: without a usable HIP device every compute call fails with
 * NABO_E_NODEVICE.
 */
#ifndef NABO_KNN_H
#define NABO_KNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NABO_METRIC_EUCLIDEAN    0   /* nabo/_mapping.py:16-26, selected when intra_ref (:119-121) */
#define NABO_METRIC_MOD_CANBERRA 1   /* nabo/_mapping.py:29-45, selected otherwise      (:122-124) */
#define NABO_METRIC_COSINE       2   /* extension (no reference counterpart): Euclidean filter on unit rows */

#define NABO_OK            0
#define NABO_E_INVALID    -1   /* bad argument (the Python shim raises ValueError)            */
#define NABO_E_NODEVICE   -2   /* no HIP device / device index out of range                   */
#define NABO_E_HIP        -3   /* a HIP runtime call failed                                   */
#define NABO_E_NOMEM      -4   /* device or host allocation failed                            */
#define NABO_E_UNSUPPORTED -5  /* size outside what an entry point can address (see each one)  */
#define NABO_E_COMM       -6   /* a collective failed or timed out, or a peer rank reported an
                                  error: see "failure semantics" at nabo_sharded_query          */

/* Limits of the instantiated FILTER kernels -- not of the API: the reference accepts any k and use_comps
 * (nabo/_mapping.py:495-524), and so do nabo_knn / nabo_index_query.  Beyond these limits every row is answered
 * by the exact float64 kernels (same results, brute-force speed).  nabo_index_query_candidates (shard mode)
 * needs g <= NABO_MAX_COMPS.  Global reference indices must stay below 2^32 - 1
 * (ref_index_base + n_ref <= 0xFFFFFFFE, checked by nabo_index_create). */
#define NABO_MAX_COMPS      128  /* use_comps (g) for the Euclidean / cosine MFMA kernel      */
#define NABO_MAX_K           56  /* k + drop_first (candidate lists hold 32 or 64 entries)     */

const char *nabo_version(void);
const char *nabo_last_error(void);
/* Number of visible HIP devices (0 when there is none; never fails). */
int nabo_device_count(void);

/* ---- coarse seam: replaces the tile loop + mask + sort of _calc_dist -------------------
 * (nabo/_mapping.py:98-146), array-in / array-out, HOST pointers, synchronous.
 * X [m,g] targets, Y [n,g] references, row-major float64 (the `[:use_comps]` prefixes the
 * reference gathers at :105,:113).  ref_mask: n bytes, non-zero = ignored reference
 * (ignore_ref_cells, :135-138), may be NULL.  Requires 1 <= k, k + drop_first <= n. */
int nabo_knn(const double *X, int64_t m, const double *Y, int64_t n, int32_t g,
             int32_t k, int32_t metric, double dist_factor,
             const uint8_t *ref_mask, int32_t drop_first,
             int64_t *out_idx, double *out_dist, int32_t device);

/* ---- fine seam: literal a1/a2 kernels (nabo/_mapping.py:16-45, call sites :120-124) -----
 * D [m,n] float64, caller-allocated host buffers (the reference's caller allocates d, :117). */
int nabo_pairwise(const double *X, int64_t m, const double *Y, int64_t n, int32_t g,
                  int32_t metric, double dist_factor, double *D, int32_t device);

/* ---- resident API: references stay in HBM across many target batches -------------------
 * One nabo_index per (device, reference shard).  ref_index_base is added to every
 * returned index (reference-row sharding across GPUs: shard r holds rows
 * [base_r, base_r + n_ref) of the global reference and reports GLOBAL indices). */
typedef struct nabo_index nabo_index;

int nabo_index_create(nabo_index **out, int32_t device, int64_t n_ref, int32_t g,
                      int32_t metric, double dist_factor, int64_t ref_index_base);
int nabo_index_destroy(nabo_index *ix);
/* Tuning options of ONE index (tests drive rows down every link of the pass chain with them; A/B tools cut launches
 * differently).  Every setting returns the SAME BITS -- an option chooses how a launch is cut or which filter pass answers a
 * row, never what the answer is (the reference has one float64 path, nabo/_mapping.py:16-45).  Names: "splits" (reference
 * splits of a filter launch, 0 = cost model), "tail_split", "lkeep" (kept list entries of the first pass), "coarse_slack",
 * "cand_slack", "seeded_pass", "coarse_adapt", "wide_retry", "refine_overlap" (0 / 1: links of the pass chain), "prepass"
 * (tournament seeds, percent of the planned length; 0 = off), "merge_lists" (several lists per row are merged by their
 * filter keys before the float64 step), "one_round" (fewer column-workgroups than slots: splits chosen to fill one round of
 * workgroups), "pieces" (experiments builds: the same query cut into equal chunks of the (column, tile) space; a no-op otherwise), "l2c_geo" (0 = A, 1 = B,
 * 2 = C), "l2_r1", "split_refs_max", "cosine_centre" (takes effect at the next set_ref).  Unknown names: NABO_E_INVALID.
 * The library reads TWO environment variables, once, in nabo_index_create: NABO_L2_MODE = f32 | f16x3 (which Euclidean /
 * cosine filter runs first; default: the one-product pass) and NABO_CANBERRA_MODE = exact | swar | bits; the sharded
 * transport reads NABO_COMM_TIMEOUT_S and NABO_RCCL_LIB. */
int nabo_index_set_option(nabo_index *ix, const char *name, int64_t value);

/* Upload / adopt the reference rows.  Y is [n_ref,g] float64 row-major; when
 * y_on_device != 0 it is a device pointer on the index's device and is BORROWED (must
 * outlive the index); otherwise it is copied.  ref_mask is a HOST pointer (n_ref bytes) or NULL. */
int nabo_index_set_ref(nabo_index *ix, const double *Y, int32_t y_on_device,
                       const uint8_t *ref_mask);

/* Replace the ignore mask of the resident references (n_ref bytes, HOST pointer, NULL = none) without
 * touching them: the graph-repair step queries "nearest reference among an allowed set" for one component
 * after another (nabo/_mapping.py:203-249). */
int nabo_index_set_mask(nabo_index *ix, const uint8_t *ref_mask);

/* k-NN of m target rows against the resident references.
 * x_on_device / out_on_device select host or device pointers for X and for
 * out_idx[m,k] (int64) / out_dist[m,k] (float64).  Work is enqueued on the index's own
 * HIP stream and the call returns after that stream has drained (results are final). */
int nabo_index_query(nabo_index *ix, const double *X, int32_t x_on_device, int64_t m,
                     int32_t k, int32_t drop_first,
                     int64_t *out_idx, double *out_dist, int32_t out_on_device);
/* The same query WITHOUT waiting for it (an asynchronous form).  nabo_index_query_async returns as soon as the
 * query has been handed to a host thread of its own (a query synchronises its stream between its passes -- the fail count
 * of one pass sizes the next -- so "enqueue and return" needs that thread); nabo_index_query_wait returns the query's
 * status (NABO_OK when nothing is in flight).  One query in flight per index; every other call on the index refuses with
 * NABO_E_INVALID until the wait (nabo_index_destroy waits itself); X and the outputs must stay valid until then.  What it is
 * for: the caller's thread goes on -- uploads the next target batch, runs the exchange of a sharded query's other half,
 * drives another index on the same or another device (kernels of two indices overlap on one GPU as far as they leave
 * each other room). */
int nabo_index_query_async(nabo_index *ix, const double *X, int32_t x_on_device, int64_t m, int32_t k,
                           int32_t drop_first, int64_t *out_idx, double *out_dist, int32_t out_on_device);
int nabo_index_query_wait(nabo_index *ix);

/* Shard mode (reference rows sharded over GPUs with GLOBAL certification, nabo_sharded_query below): the first
 * n_cand (<= 32) entries of the shard's order rows WITHOUT a local verdict -- out_idx [m,n_cand] (global
 * indices, -1 = absent), out_dist [m,n_cand] (exact float64, +inf = absent) -- and out_bound [m]: a lower
 * bound on the exact SQUARED distance of every reference of this shard that is not in the emitted list
 * (+inf: nothing else exists; -inf: unknown, the caller must fall back to nabo_index_query).  Euclidean and
 * cosine (there the bound is the SQUARE of a lower bound on the cosine distance, so the same test applies);
 * device pointers for the three outputs.  A merged k'-th distance d with d^2 < min over shards of
 * out_bound is the exact global k'-th distance. */
int nabo_index_query_candidates(nabo_index *ix, const double *X, int32_t x_on_device, int64_t m,
                                int32_t n_cand, int64_t *out_idx, double *out_dist, double *out_bound);

/* HIP-event timings (ms) and counters of the LAST nabo_index_query on this index.
 * ms[0] pack targets, ms[1] distance+top-k kernel (the dominant kernel), ms[2] float64
 * refine, ms[3] exact fallback for guard-flagged rows, ms[4] total on-stream.
 * counters[0] rows re-solved by the exact fallback, counters[1] candidate lists per row (S),
 * counters[2] candidates per list (L), counters[3] workgroups of the dominant kernel. */
int nabo_index_last_stats(const nabo_index *ix, double ms[5], int64_t counters[4]);
/* Name of the dominant kernel the LAST query on this index ran (NUL-terminated into buf[n]): which filter the
 * launch logic picked -- the one-product pass on the f16 matrix pipe (default where instantiated: g < 64 and
 * k + drop_first <= 28), the f16x3 split (NABO_L2_MODE=f16x3 pins it as the first pass), the fp32-MFMA kernel
 * (NABO_L2_MODE=f32), the Canberra filter, or the exact kernels. */
int nabo_index_last_kernel(const nabo_index *ix, char *buf, size_t n);
/* Rows of the LAST query that a filter pass could not certify, by the pass they went on to: rows[0] the SEEDED
 * one-product pass (every row starts from the threshold its failed certificate implies), rows[1] the f16x3 pass, rows[2]
 * the 64-entry lists; what is left after all of them is counters[0] of nabo_index_last_stats (exact float64 kernels).
 * The default Euclidean / cosine filter for g < 64 and k + drop_first <= 28 starts with the one-product pass;
 * NABO_L2_MODE=f16x3 makes the f16x3 filter the first pass.  Results are the same bits whichever pass answers a row
 * (the reference has one float64 path: nabo/_mapping.py:16-26). */
int nabo_index_last_passes(const nabo_index *ix, int64_t rows[3]);
/* WHICH pass answered each of the m rows of the LAST nabo_index_query on this index (out [m] bytes, host; m must be that
 * query's row count): a test can then compare exactly the rows that took an unusual route with the reference's
 * nabo/_mapping.py:139-145, instead of hoping that a uniform sample contains some. */
#define NABO_PASS_ONE_PRODUCT 0   /* the one-product f16 filter, lists built from +inf (the default first pass)      */
#define NABO_PASS_SEEDED      1   /* the same filter, seeded with the threshold the row's failed certificate implies */
#define NABO_PASS_SECOND      2   /* the f16x3 split or the fp32-MFMA filter (first pass under NABO_L2_MODE=f16x3|f32) */
#define NABO_PASS_WIDE        3   /* the 64-entry lists (second chance of rows the 32-entry lists could not certify)  */
#define NABO_PASS_EXACT       4   /* the exact float64 kernels (brute force)                                          */
#define NABO_PASS_CANBERRA    5   /* the modified-Canberra filter (count + fp32 lower bound + float64 refine)         */
int nabo_index_last_row_pass(const nabo_index *ix, uint8_t *out, int64_t m);

/* The launch plan of a Euclidean / cosine query WITHOUT an index or a device: what nabo_index_query (n_cand = 0) or
 * nabo_index_query_candidates (n_cand > 0) would launch first for this shape on a part with n_cu compute units -- a pure
 * function of its arguments (l2_mode: what NABO_L2_MODE would hold, NULL = default; options: "name=value,..." of
 * nabo_index_set_option, NULL = defaults).  out: [0] NABO_PASS_* of the first filter, [1] geometry of the one-product kernel
 * (-1: another kernel), [2] target rows per workgroup, [3] / [4] workgroups (x) of the main / tail launch, [5] / [6] their
 * reference splits, [7] kept list entries, [8] emitted list length, [9] reference tiles per split, [10] / [11] tournament
 * tiles and tiles per group (0: no tournament), [12] workgroups resident at once, [13] workgroups launched in all,
 * [14] padded target rows, [15] operand steps of 16 slots, [16] 1 when the launch is cut into PIECES (fewer column-
 * workgroups than slots: [13] workgroups work through equal chunks of the (column, reference tile) space, [5] = lists per
 * row, [9] = the longest a piece can be), [17] tiles per chunk.  kernel (optional): the kernel's name as nabo_index_last_kernel
 * reports it. */
#define NABO_PLAN_FIELDS 18
int nabo_query_plan(int64_t n_ref, int32_t g, int32_t metric, int64_t m, int32_t k, int32_t drop_first, int32_t n_cand,
                    int32_t n_cu, const char *l2_mode, const char *options, int64_t out[NABO_PLAN_FIELDS], char *kernel,
                    size_t kernel_len);

/* ---- shard merge (reference rows sharded over GPUs) ---------------
 * parts_idx / parts_dist: [n_parts, m, kp] DEVICE arrays, each row sorted by the canonical
 * order with GLOBAL indices (what nabo_index_query(k=kp, drop_first=0) returns on every
 * shard after the RCCL exchange).  Writes the merged first k entries after the optional
 * positional drop to out_idx/out_dist [m,k] (device).  Masked references must have been
 * excluded by the shards (entries with idx < 0 are treated as absent). */
int nabo_merge_topk(int32_t device, const int64_t *parts_idx, const double *parts_dist,
                    int32_t n_parts, int64_t m, int32_t kp, int32_t k, int32_t drop_first,
                    int64_t *out_idx, double *out_dist);

/* ---- reference rows sharded over the GPUs of one node ------------------------------------
 * The reference has no multi-device path; the call site served is Mapping.calc_dist (nabo/_mapping.py:441-444).
 * Rank r of N holds reference rows [base_r, base_r + n_r) in its own nabo_index (ref_index_base = base_r; a shard
 * with fewer than k + drop_first rows takes part with what it has) and sees ALL m target rows; after the call every
 * rank holds the full [m,k] result, which equals the unsharded index bit for bit.  Transport: RCCL over xGMI (librccl.so is loaded on first use); no torch,
 * no MPI.  A communicator owns a HIP stream and is used by one host thread at a time.
 *
 *   one process per GPU:    rank 0 calls nabo_comm_unique_id, hands the NABO_COMM_ID_BYTES bytes to the other ranks by
 *                           any means (file, socket, launcher), every rank calls nabo_comm_create (collective);
 *   one process, n devices: nabo_comm_create_all fills comms[n] (ncclCommInitAll); the caller drives every rank from
 *                           its own host thread (collectives block until all ranks have entered);
 *   loopback:               n ranks as host threads of ONE process exchanging through device-to-device copies instead
 *                           of RCCL -- devices may repeat, so N shards can be run on a single GPU (tests, rehearsal).
 */
#define NABO_COMM_ID_BYTES 128
typedef struct nabo_comm nabo_comm;

int nabo_comm_unique_id(void *id /* NABO_COMM_ID_BYTES */);
int nabo_comm_create(nabo_comm **out, int32_t device, int32_t rank, int32_t world, const void *id);
int nabo_comm_create_all(nabo_comm **comms /* [n] */, const int32_t *devices, int32_t n);
int nabo_comm_create_loopback(nabo_comm **comms /* [n] */, const int32_t *devices, int32_t n);
int nabo_comm_destroy(nabo_comm *c);
int nabo_comm_rank(const nabo_comm *c);
int nabo_comm_world(const nabo_comm *c);
/* How many ranks the TRANSPORT itself says the communicator has -- ncclCommCount for RCCL, the rendezvous' size for the
 * loopback transport -- as opposed to what the caller asked for (nabo_comm_world): a benchmark line reports this one. */
int nabo_comm_transport_ranks(nabo_comm *c);
/* Give up on a communicator from ANY thread: every rank blocked in one of its collectives (and every later call on
 * it) returns NABO_E_COMM -- ncclCommAbort for RCCL, the rendezvous' abort flag for the loopback transport.  The
 * handle must still be destroyed.  nabo_comm_set_timeout: how long a rank waits for its peers inside a collective
 * before it aborts the communicator itself (seconds; default 600, or NABO_COMM_TIMEOUT_S at creation). */
int nabo_comm_abort(nabo_comm *c);
int nabo_comm_set_timeout(nabo_comm *c, double seconds);
/* Collective helpers for a host that has no other communication layer (bench.py's timing bracket):
 * barrier, and MAX over ranks of one non-negative host double (in place). */
/* 2-D layout for the sharded query (optional; 0 or the world size = the 1-D form): the references are cut into
 * ref_shards pieces, rank r holds piece r % ref_shards (the caller builds its index from that piece) and answers for target
 * slice r / ref_shards; the exchange, merge and certificate run inside each group of ref_shards ranks, the final gather
 * over all ranks.  A shard's list work does not shrink with the shard (every row fills a list on every piece): fewer,
 * larger pieces cost less of it.  ref_shards must divide the world size; global-certification protocol only. */
int nabo_comm_set_ref_shards(nabo_comm *c, int32_t ref_shards);
int nabo_comm_barrier(nabo_comm *c);
int nabo_comm_allreduce_max_f64(nabo_comm *c, double *value);

/* Entries each shard emits under global certification: the smallest list length that leaves an expected < 0.1 rows
 * of an m-row batch for the second round, world * m * P[Bin(kk, 1/world) >= Ls] < 0.1, within [ceil(kk/world), kk+1]
 * and <= 32.  kk = k + drop_first. */
int32_t nabo_candidates_per_shard(int32_t kk, int32_t world, int64_t m);

/* The sharded query (collective: every rank calls it with the same m, k, drop_first, protocol and the same X).
 * X [m,g], out_idx [m,k], out_dist [m,k]: DEVICE pointers on the communicator's device.
 * protocol 0 = automatic; 1 = global certification (Euclidean / cosine: every shard emits
 * nabo_candidates_per_shard entries + a bound on everything else, ONE grouped exchange to the owner of each target
 * row, merge, the owner accepts a row when its k'-th distance lies below every shard's bound; rows it refuses are
 * re-solved exactly in a second, small round); 2 = local certification (every shard's certified first k' entries;
 * the only form for the modified Canberra metric).  The positional drop (nabo/_mapping.py:142) is applied after
 * the merge.  Returns after the communicator's stream has drained.
 * Absent entries: index -1 (the distance beside it is NaN in merged results and +inf in candidate lists -- test the
 * index).  Ignored references (ref_mask): with more than one shard a row with fewer than k' unmasked references in
 * the WHOLE reference set ends in absent entries; the one-device path continues such a row with the ignored
 * references by index, as numpy.ma's NaN fill does (nabo/_mapping.py:135-146).
 * Failure semantics (the reference is a single process, nabo/_mapping.py:48-148 -- nothing to match; the rule is
 * that no rank waits for a peer that has given up): what a rank can get wrong ALONE -- its arguments (every rank must
 * pass the same m, k, drop_first, protocol: checked), a buffer it cannot allocate, its local queries -- is agreed on
 * by all ranks before anything is exchanged: then EVERY rank returns an error (the failing rank its own status and
 * message, the others NABO_E_COMM) and the communicator stays usable.  An error inside a collective (RCCL failure, a
 * peer that never arrives within the timeout, nabo_comm_abort) aborts the communicator: every rank returns
 * NABO_E_COMM and so does every later call on it. */
int nabo_sharded_query(nabo_comm *c, nabo_index *ix, const double *X, int64_t m, int32_t k, int32_t drop_first,
                       int64_t *out_idx, double *out_dist, int32_t protocol);
/* ms: [0] local query (this rank's shard), [1] exchange, [2] merge + certificate, [3] second round, [4] slice,
 * [5] final all-gather, [6] total on the communicator's stream, [7] the distance + top-k kernel of [0].  counters: [0] rows re-solved in the second round
 * (all ranks), [1] candidates per shard (0 under local certification), [2] unused, [3] protocol used (1 / 2). */
int nabo_sharded_last_stats(const nabo_comm *c, double ms[8], int64_t counters[4]);

/* nabo_knn with the reference rows sharded over several GPUs of ONE node, for a caller that has no threads, communicators
 * or device buffers of its own (the devices[] form of the boundary; the call site is Mapping.calc_dist,
 * nabo/_mapping.py:408-444 -> _calc_dist :48-148).  Host arrays in, host arrays out, same results as nabo_knn on one device
 * (N shards = 1 shard bit for bit; rows with fewer than k + drop_first unmasked references in the whole set: see
 * nabo_sharded_query).  Internally one host thread per device: rank r indexes reference rows [n r / N, n (r + 1) / N),
 * uploads all m target rows, and the ranks meet in nabo_sharded_query.  transport 0 = RCCL (devices must differ),
 * 1 = loopback (devices may repeat: the whole protocol on one GPU, rehearsal and tests).  n_devices = 1: nabo_knn.
 * A rank that fails alone fails the call with its status and message; a collective failure: NABO_E_COMM. */
int nabo_knn_devices(const double *X, int64_t m, const double *Y, int64_t n, int32_t g, int32_t k, int32_t metric,
                     double dist_factor, const uint8_t *ref_mask, int32_t drop_first, const int32_t *devices,
                     int32_t n_devices, int32_t transport, int64_t *out_idx, double *out_dist);

/* ---- SNN edge counts on device (consumer of the top-k: nabo/_mapping.py:186-198) --------
 * t_idx [m,k], r_idx [n,k] int64 DEVICE arrays (first k of the order rows).  For every
 * (t, slot s) writes out_snn[t*k+s] = | set(t_idx[t]) & set(r_idx[t_idx[t,s]]) | (int32,
 * device).  The weight round(snn/(2(k-1)-snn),2) and the snn>0 filter stay with the caller. */
int nabo_snn_counts(int32_t device, const int64_t *t_idx, int64_t m,
                    const int64_t *r_idx, int64_t n, int32_t k, int32_t *out_snn);

/* ---- host-side graph assembly (consumers of the top-k; plain C++ on the caller's cores, no GPU needed) ------------
 * nabo_pyset_order: rows [n,k] int64 of DISTINCT non-negative ints (HOST) -> perm [n,k] int32: the column order in which
 * CPython iterates set(row).  nabo/_mapping.py:190-191 walks a cell's neighbours in that order and networkx keeps
 * insertion order, so it is the row order of every node's dataset in the `<uid>_graph` groups (nabo/_mapping.py:252-273). */
int nabo_pyset_order(const int64_t *rows, int64_t n, int32_t k, int32_t *perm);
/* Connected components of an undirected edge list a[e] -- b[e] over nodes 0..n-1 (nabo/_mapping.py:203-214:
 * nx.connected_components): labels[i] = smallest node index of i's component.  HOST pointers. */
int nabo_component_labels(int64_t n, const int64_t *a, const int64_t *b, int64_t n_edges, int64_t *labels);
/* Adjacency rows per node from (node, neighbour, weight) rows in insertion order, as networkx's dict-of-dicts keeps them
 * and nabo/_mapping.py:252-273 dumps them: a node's neighbours in the order they were first added, a repeated (node,
 * neighbour) pair keeps its first position and takes its LAST weight.  starts [n_nodes+1], nbr_out / w_out [n_rows]
 * (the first starts[n_nodes] entries are used).  HOST pointers; node in [0, n_nodes), neighbour >= 0. */
int nabo_group_edges(int64_t n_nodes, int64_t n_rows, const int64_t *node, const int64_t *nbr, const double *w,
                     int64_t *starts, int64_t *nbr_out, double *w_out);

/* ---- permutation null for mapping scores (EXTENSION: BASELINE.json configs[4]; the reference has the
 * score, Graph.get_mapping_score nabo/_graph.py:555-697, but no permutation test) -------------------------
 * Bipartite target->reference edges in CSR by reference node: row_ptr [n_ref+1], edge_t [E] (pooled target
 * cell of the edge), edge_w [E]; group [n_t] flags the sample of interest (n_A cells).  HOST pointers.
 *   out_obs[r]  = multiplier * sum_{e in row r, group[t_e]} w_e / n_A            (the reference's score)
 *   permutation p: key(t,p) = top key_bits of splitmix64(seed + (p+1)*0x9E3779B97F4A7C15 + t*0xD1B54A32D192ED03),
 *                  label_p[t] = key <= (n_A-th smallest key); out_sizes[p] = #labelled (n_A unless keys tie)
 *   out_nge[r]  = #{p : multiplier * sum_{e in row r, label_p[t_e]} w_e / out_sizes[p]  >=  out_obs[r]}
 *   out_mean/out_sd[r] = mean and population sd of the permuted scores.  Edge sums are float64, in row order.
 * key_bits in {8,16,..,64} (64 in production; small values force ties, for tests); n_perm <= 4096. */
int nabo_score_null(int32_t device, int64_t n_ref, const int64_t *row_ptr, const int64_t *edge_t,
                    const double *edge_w, int64_t n_t, const uint8_t *group, int32_t n_perm, uint64_t seed,
                    int32_t key_bits, double multiplier, double *out_obs, int64_t *out_nge, double *out_mean,
                    double *out_sd, int64_t *out_sizes /* [n_perm] or NULL */);

/* The same null from an edge LIST in any order (what a mapping yields: one (reference cell, target cell, weight)
 * triple per kept neighbour): the CSR is built on the device by a stable sort on edge_ref, so a reference
 * node's edges are summed in the order the caller listed them -- results are identical to nabo_score_null on
 * the CSR a stable host sort would give.  HOST pointers; n_edges < 2^32-1. */
int nabo_score_null_edges(int32_t device, int64_t n_ref, int64_t n_edges, const int64_t *edge_ref,
                          const int64_t *edge_t, const double *edge_w, int64_t n_t, const uint8_t *group,
                          int32_t n_perm, uint64_t seed, int32_t key_bits, double multiplier, double *out_obs,
                          int64_t *out_nge, double *out_mean, double *out_sd, int64_t *out_sizes);

/* ---- plain device-memory helpers so a ctypes host needs no other GPU binding ------------ */
int nabo_dev_malloc(int32_t device, void **ptr, size_t bytes);
int nabo_dev_free(int32_t device, void *ptr);
int nabo_memcpy_h2d(int32_t device, void *dst, const void *src, size_t bytes);
int nabo_memcpy_d2h(int32_t device, void *dst, const void *src, size_t bytes);
int nabo_dev_synchronize(int32_t device);
/* free / total bytes of the device's memory (hipMemGetInfo): what a long-running host watches for leaks */
int nabo_dev_mem_info(int32_t device, size_t *free_bytes, size_t *total_bytes);

#ifdef __cplusplus
}
#endif
#endif /* NABO_KNN_H */
