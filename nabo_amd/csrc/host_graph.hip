// host_graph.hip -- HOST-side graph assembly behind the C ABI (include/nabo_knn.h): what the reference does in Python
// between its distance kernels and its HDF5 file, once the top-k lists are there.  No device code, no HIP call: plain C++
// on the caller's cores (std::thread), so these entry points work without a GPU too.
//   nabo_pyset_order        the order in which CPython iterates set(row) (nabo/_mapping.py:190-191 walks a cell's
//                           neighbours that way, networkx keeps the insertion order: it is the row order of every node's
//                           dataset in the <uid>_graph groups)
//   nabo_component_labels   connected components of an edge list (nabo/_mapping.py:203-214: nx.connected_components)
//   nabo_group_edges        adjacency rows per node in insertion order, a repeated (node, neighbour) pair keeps its first
//                           position and its last weight (networkx's dict-of-dicts; nabo/_mapping.py:252-273 dumps it)
// The numpy forms these replace took 3.1 + 2.0 + 2.8 s of a 1M-cell Mapping run.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <exception>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/nabo_knn.h"

namespace nabo {
int api_fail(int code, const char *fmt, ...);

namespace {

// Rows [0, n) in chunks on the caller's cores.  Nothing escapes a worker thread: an exception thrown inside one (bad_alloc
// from a per-thread table) is kept, every thread is joined, and the FIRST exception is rethrown on the calling thread, where
// the extern "C" entry points turn it into a status code; if a thread cannot be started (system_error) the threads already
// running are joined first and the rest of the rows is done by the caller.
template <typename F>
void parallel_rows(int64_t n, int64_t grain, F &&f)
{
    if (n <= 0) return;
    unsigned hw = std::thread::hardware_concurrency();
    int64_t nt = hw ? (int64_t)hw : 4;
    if (nt > 32) nt = 32;
    if (nt > (n + grain - 1) / grain) nt = (n + grain - 1) / grain;
    if (nt <= 1) { f(0, n); return; }
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err((size_t)nt);
    const int64_t per = (n + nt - 1) / nt;
    int64_t done_to = 0;                                     // rows [0, done_to) have a thread
    try {
        th.reserve((size_t)nt);
        for (int64_t t = 0; t < nt; ++t) {
            const int64_t a = t * per, b = std::min(n, a + per);
            if (a >= b) break;
            std::exception_ptr *slot = &err[(size_t)t];
            th.emplace_back([&f, a, b, slot]() {
                try { f(a, b); } catch (...) { *slot = std::current_exception(); }
            });
            done_to = b;
        }
    } catch (...) {                                          // thread creation failed: the caller's thread takes the rest
        done_to = th.empty() ? 0 : std::min(n, (int64_t)th.size() * per);
    }
    std::exception_ptr mine;
    if (done_to < n) {
        try { f(done_to, n); } catch (...) { mine = std::current_exception(); }
    }
    for (auto &x : th) x.join();
    for (auto &e : err)
        if (e) std::rethrow_exception(e);
    if (mine) std::rethrow_exception(mine);
}

// CPython's set of small non-negative ints (Objects/setobject.c, 3.7 .. 3.12): open addressing, hash(int) = int, a table
// of 8 slots, 9 linear probes behind the first slot when they fit below the mask, then i = 5 i + 1 + (perturb >>= 5);
// after an insertion that brings fill * 5 to mask * 3 the table is rebuilt with the first power of two above 4 * used
// (2 * used beyond 50 000 entries), old entries re-inserted in table order.  Keys of a row are distinct: no equality tests.
struct PySet {
    std::vector<int32_t> table, old;      // column ids, -1 = empty
    int64_t mask = 7;
    void place(const int64_t *row, int32_t col)
    {
        const uint64_t h = (uint64_t)row[col];
        uint64_t perturb = h;
        uint64_t i = h & (uint64_t)mask;
        for (;;) {
            if (table[i] < 0) { table[i] = col; return; }
            if (i + 9 <= (uint64_t)mask)
                for (int j = 1; j <= 9; ++j)
                    if (table[i + j] < 0) { table[i + j] = col; return; }
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & (uint64_t)mask;
        }
    }
    void order(const int64_t *row, int k, int32_t *out)
    {
        mask = 7;
        table.assign(8, -1);
        for (int c = 0; c < k; ++c) {
            place(row, c);
            const int64_t fill = c + 1;
            if (fill * 5 >= mask * 3) {
                const int64_t minused = fill > 50000 ? fill * 2 : fill * 4;
                int64_t newsize = 8;
                while (newsize <= minused) newsize <<= 1;
                old.swap(table);
                table.assign((size_t)newsize, -1);
                mask = newsize - 1;
                for (int32_t cc : old)
                    if (cc >= 0) place(row, cc);
            }
        }
        int o = 0;
        for (int32_t cc : table)
            if (cc >= 0) out[o++] = cc;
    }
};

}  // namespace
}  // namespace nabo

using namespace nabo;

extern "C" {

int nabo_pyset_order(const int64_t *rows, int64_t n, int32_t k, int32_t *perm)
{
    try {
        if (n < 0 || k < 0 || (n > 0 && k > 0 && (!rows || !perm))) return api_fail(NABO_E_INVALID, "nabo_pyset_order: bad arguments");
        if (n == 0 || k == 0) return NABO_OK;
        std::atomic<int> bad{0};
        parallel_rows(n, 4096, [&](int64_t a, int64_t b) {
            PySet s;
            for (int64_t r = a; r < b; ++r) {
                const int64_t *row = rows + r * k;
                for (int c = 0; c < k; ++c)
                    if (row[c] < 0) { bad = 1; return; }
                s.order(row, k, perm + r * k);
            }
        });
        if (bad) return api_fail(NABO_E_INVALID, "ERROR: neighbour indices must be non-negative");
        return NABO_OK;
    } catch (const std::bad_alloc &) {
        return api_fail(NABO_E_NOMEM, "nabo_pyset_order: out of host memory");
    } catch (const std::system_error &e) {
        return api_fail(NABO_E_INVALID, "nabo_pyset_order: %s", e.what());
    }
}

int nabo_component_labels(int64_t n, const int64_t *a, const int64_t *b, int64_t n_edges, int64_t *labels)
{
    try {
        if (n < 0 || n_edges < 0 || (n > 0 && !labels) || (n_edges > 0 && (!a || !b)))
            return api_fail(NABO_E_INVALID, "nabo_component_labels: bad arguments");
        for (int64_t i = 0; i < n; ++i) labels[i] = i;
        // union-find, the larger root hooked under the smaller one: a root is the smallest member of its tree
        auto find = [labels](int64_t x) {
            while (labels[x] != x) {
                labels[x] = labels[labels[x]];
                x = labels[x];
            }
            return x;
        };
        for (int64_t e = 0; e < n_edges; ++e) {
            if (a[e] < 0 || a[e] >= n || b[e] < 0 || b[e] >= n) return api_fail(NABO_E_INVALID, "nabo_component_labels: node out of range");
            const int64_t ra = find(a[e]), rb = find(b[e]);
            if (ra < rb) labels[rb] = ra;
            else if (rb < ra) labels[ra] = rb;
        }
        for (int64_t i = 0; i < n; ++i) labels[i] = find(i);
        return NABO_OK;
    } catch (const std::bad_alloc &) {
        return api_fail(NABO_E_NOMEM, "nabo_component_labels: out of host memory");
    } catch (const std::system_error &e) {
        return api_fail(NABO_E_INVALID, "nabo_component_labels: %s", e.what());
    }
}

int nabo_group_edges(int64_t n_nodes, int64_t n_rows, const int64_t *node, const int64_t *nbr, const double *w,
                     int64_t *starts, int64_t *nbr_out, double *w_out)
{
    try {
        if (n_nodes < 0 || n_rows < 0 || !starts || (n_rows > 0 && (!node || !nbr || !w || !nbr_out || !w_out)))
            return api_fail(NABO_E_INVALID, "nabo_group_edges: bad arguments");
        if (n_nodes == 0 || n_rows == 0) {
            if (n_rows > 0) return api_fail(NABO_E_INVALID, "nabo_group_edges: node out of range");
            for (int64_t i = 0; i <= n_nodes; ++i) starts[i] = 0;
            return NABO_OK;
        }
        // A stable counting sort by node in two levels, so that every scatter writes into a cache-sized window: rows go to
        // NBK node ranges first (each thread scatters its chunk of the rows; chunk order inside a range = insertion order), then
        // every range sorts its rows by node on its own (threads over ranges).  30M rows: 0.5 s where one flat scatter took 3.
        constexpr int64_t NBK = 1024;
        const int64_t span = (n_nodes + NBK - 1) / NBK;                 // nodes per range
        unsigned hw = std::thread::hardware_concurrency();
        int64_t T = hw ? (int64_t)hw : 4;
        if (T > 32) T = 32;
        if (T > (n_rows + 65535) / 65536) T = (n_rows + 65535) / 65536;
        const int64_t per = (n_rows + T - 1) / T;
        std::vector<int64_t> hist((size_t)(T * NBK), 0);
        std::atomic<int> bad{0};
        parallel_rows(T, 1, [&](int64_t t0, int64_t t1) {
            for (int64_t t = t0; t < t1; ++t) {
                int64_t *h = hist.data() + t * NBK;
                for (int64_t r = t * per, e = std::min(n_rows, r + per); r < e; ++r) {
                    if (node[r] < 0 || node[r] >= n_nodes || nbr[r] < 0) { bad = 1; return; }
                    ++h[node[r] / span];
                }
            }
        });
        if (bad) return api_fail(NABO_E_INVALID, "nabo_group_edges: node out of range");
        std::vector<int64_t> bstart((size_t)NBK + 1, 0);
        {
            int64_t run = 0;
            for (int64_t bk = 0; bk < NBK; ++bk) {
                bstart[(size_t)bk] = run;
                for (int64_t t = 0; t < T; ++t) {
                    const int64_t c = hist[(size_t)(t * NBK + bk)];
                    hist[(size_t)(t * NBK + bk)] = run;                  // where chunk t writes its rows of range bk
                    run += c;
                }
            }
            bstart[(size_t)NBK] = run;
        }
        std::vector<int64_t> pn((size_t)n_rows), pb((size_t)n_rows);
        std::vector<double> pw((size_t)n_rows);
        parallel_rows(T, 1, [&](int64_t t0, int64_t t1) {
            for (int64_t t = t0; t < t1; ++t) {
                int64_t *h = hist.data() + t * NBK;
                for (int64_t r = t * per, e = std::min(n_rows, r + per); r < e; ++r) {
                    const int64_t p = h[node[r] / span]++;
                    pn[(size_t)p] = node[r];
                    pb[(size_t)p] = nbr[r];
                    pw[(size_t)p] = w[r];
                }
            }
        });
        // per range: rows by node (stable), then, node by node, a repeated neighbour keeps its first position and takes the last
        // weight (rows of a node are few: a sorted copy finds the repeats); kept rows are compacted in place
        std::vector<int64_t> first((size_t)n_nodes + 1, 0), kept((size_t)n_nodes, 0);
        std::vector<int64_t> tn((size_t)n_rows);
        std::vector<double> tw((size_t)n_rows);
        parallel_rows(NBK, 1, [&](int64_t b0, int64_t b1) {
            std::vector<int64_t> cur;
            std::vector<std::pair<int64_t, int64_t>> key;                // (neighbour, position)
            for (int64_t bk = b0; bk < b1; ++bk) {
                const int64_t lo = bk * span, hi = std::min(n_nodes, lo + span);
                if (lo >= hi) continue;
                const int64_t rs = bstart[(size_t)bk], re = bstart[(size_t)bk + 1];
                cur.assign((size_t)(hi - lo) + 1, 0);
                for (int64_t r = rs; r < re; ++r) ++cur[(size_t)(pn[(size_t)r] - lo) + 1];
                int64_t run = rs;
                for (int64_t i = lo; i < hi; ++i) {
                    const int64_t c = cur[(size_t)(i - lo) + 1];
                    first[(size_t)i] = run;
                    cur[(size_t)(i - lo)] = run;
                    run += c;
                }
                for (int64_t r = rs; r < re; ++r) {
                    const int64_t p = cur[(size_t)(pn[(size_t)r] - lo)]++;
                    tn[(size_t)p] = pb[(size_t)r];
                    tw[(size_t)p] = pw[(size_t)r];
                }
                for (int64_t i = lo; i < hi; ++i) {
                    const int64_t s = first[(size_t)i], e = i + 1 < hi ? first[(size_t)i + 1] : re;
                    int64_t c = e - s;
                    if (c > 1) {
                        key.clear();
                        for (int64_t p = s; p < e; ++p) key.emplace_back(tn[(size_t)p], p);
                        std::sort(key.begin(), key.end());
                        bool dup = false;
                        size_t f = 0;                                    // first row of the current run of equal neighbours
                        for (size_t q = 1; q < key.size(); ++q) {
                            if (key[q].first != key[q - 1].first) { f = q; continue; }
                            dup = true;
                            tw[(size_t)key[f].second] = tw[(size_t)key[q].second];
                            tn[(size_t)key[q].second] = -1;
                        }
                        if (dup) {
                            int64_t o = s;
                            for (int64_t p = s; p < e; ++p)
                                if (tn[(size_t)p] >= 0) { tn[(size_t)o] = tn[(size_t)p]; tw[(size_t)o] = tw[(size_t)p]; ++o; }
                            c = o - s;
                        }
                    }
                    kept[(size_t)i] = c;
                }
            }
        });
        first[(size_t)n_nodes] = n_rows;
        starts[0] = 0;
        for (int64_t i = 0; i < n_nodes; ++i) starts[i + 1] = starts[i] + kept[(size_t)i];
        parallel_rows(n_nodes, 65536, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i) {
                const int64_t c = kept[(size_t)i];
                if (c > 0) {
                    std::memcpy(nbr_out + starts[i], tn.data() + first[(size_t)i], (size_t)c * sizeof(int64_t));
                    std::memcpy(w_out + starts[i], tw.data() + first[(size_t)i], (size_t)c * sizeof(double));
                }
            }
        });
        return NABO_OK;
    } catch (const std::bad_alloc &) {
        return api_fail(NABO_E_NOMEM, "nabo_group_edges: out of host memory");
    } catch (const std::system_error &e) {
        return api_fail(NABO_E_INVALID, "nabo_group_edges: %s", e.what());
    }
}

}  // extern "C"
