"""Reference rows sharded over the GPUs of one node, torch-free (include/nabo_knn.h
`nabo_comm_*`, `nabo_sharded_query`).  See DESIGN.md section 5 for the protocol."""
from math import comb


def shard_bounds(n, world, rank):
    """Contiguous, balanced reference-row partition: rows [lo, hi) for `rank`."""
    return (n * rank) // world, (n * (rank + 1)) // world


def candidates_per_shard(kk, world, m=None):
    """Entries each shard emits in the globally certified protocol.  A row needs the second round when some shard
    holds at least Ls of its global top-k'; for exchangeable shards that is world * P[Bin(k', 1/world) >= Ls] per
    row.  With `m` (target rows) given, Ls is the smallest length that leaves an expected < 0.1 such rows in the
    whole batch -- never more than k'+1 (a shard cannot hold more than k' of the top k') -- because a longer list
    costs ~0.7 ms per entry and step while the second round costs ~3.5 ms plus three more collectives.  Without
    `m`: the share k'/N with 50 % head room, +6.  (The C side restates this rule: nabo_candidates_per_shard.)"""
    cap = min(kk + 1, 32)
    if m is None:
        ls = max(8, -(-3 * kk // (2 * world)) + 6, -(-kk // world))
        return min(ls, kk + 8, 32)
    p = 1.0 / world
    tail = 0.0
    ls = cap
    for j in range(kk, 0, -1):                       # tail = P[Bin(kk, p) >= j]
        tail += comb(kk, j) * p ** j * (1.0 - p) ** (kk - j)
        if tail * world * m >= 0.1:
            ls = j + 1
            break
        ls = j
    return max(min(ls, cap), -(-kk // world), 1)


# ---- communicators and the sharded query through the C ABI ------------------------------------------------------
import ctypes as C      # noqa: E402
import os               # noqa: E402
import queue            # noqa: E402
import threading        # noqa: E402
import time             # noqa: E402

import numpy as np      # noqa: E402

from . import _lib      # noqa: E402

ID_BYTES = 128
PROTOCOLS = {"auto": 0, "global": 1, "local": 2}


def _launcher_start_time():
    """Wall-clock time the launcher (this rank's parent process) was started at, or 0.0 when /proc does not say."""
    try:
        with open("/proc/%d/stat" % os.getppid()) as f:
            ticks = float(f.read().rsplit(")", 1)[1].split()[19])          # field 22: starttime, in clock ticks since boot
        with open("/proc/stat") as f:
            btime = next(float(ln.split()[1]) for ln in f if ln.startswith("btime"))
        return btime + ticks / os.sysconf("SC_CLK_TCK")
    except Exception:      # noqa: BLE001
        return 0.0


def id_file_path():
    """Where the ranks of ONE launch on ONE node meet: a private directory (mode 0700, per user) and a name built from
    what the launcher gives every rank identically -- MASTER_PORT, the launcher's pid and, under torchrun, the run id
    and the restart count (a restarted worker group must not pick up the id of the group that died)."""
    d = os.environ.get("NABO_ID_DIR") or os.path.join("/tmp", "nabo-%d" % os.getuid())
    os.makedirs(d, mode=0o700, exist_ok=True)
    tag = "%s_%d_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.getppid(),
                           os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))
    return os.path.join(d, "rccl_%s.id" % "".join(ch if ch.isalnum() or ch in "_-" else "-" for ch in tag))


def exchange_unique_id(rank, world, make_id, path=None, timeout=300.0, not_before=None):
    """One process per GPU on ONE node: rank 0 creates the RCCL unique id (`make_id()` -> 128 bytes) and publishes it
    through a file; the other ranks poll for it.  Rank 0 removes whatever is left under that name first, writes under
    a temporary name (O_EXCL, mode 0600) and renames: readers never see a partial or foreign file.  A reader accepts
    only a file written after `not_before` -- by default the start of the launcher process, which precedes every rank
    of this launch and follows every earlier job that could have used the same name.  Rank 0 removes the file when
    the communicator is closed.  Any other channel works as well (nabo_comm_create only needs the bytes)."""
    if path is None:
        path = id_file_path()
    if rank == 0:
        blob = bytes(make_id())
        assert len(blob) == ID_BYTES
        try:
            os.remove(path)                                    # a leftover of a crashed job
        except OSError:
            pass
        tmp = "%s.%d.tmp" % (path, os.getpid())
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(blob)
        os.replace(tmp, path)
        return blob, path
    if not_before is None:
        not_before = _launcher_start_time()
    t0 = time.time()
    while True:
        try:
            st = os.stat(path)
            if st.st_mtime >= not_before - 1.0 and st.st_uid == os.getuid():
                with open(path, "rb") as f:
                    blob = f.read()
                if len(blob) == ID_BYTES:
                    return blob, path
        except OSError:
            pass
        if time.time() - t0 > timeout:
            raise _lib.NaboError("rank %d: no RCCL unique id at %s after %.0f s" % (rank, path, timeout))
        time.sleep(0.01)


class Comm:
    """One rank's communicator (nabo_comm, include/nabo_knn.h).  No torch: RCCL is reached through libnabo_knn.so."""

    def __init__(self, handle, device, rank, world, id_path=None):
        self._h, self.device, self.rank, self.world, self._id_path = handle, device, rank, world, id_path

    @classmethod
    def from_env(cls, device=None):
        """RANK / WORLD_SIZE / LOCAL_RANK as torch.distributed.run (or any launcher) exports them."""
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        L = _lib.lib()

        def make_id():
            buf = C.create_string_buffer(ID_BYTES)
            _lib.check(L.nabo_comm_unique_id(buf))
            return buf.raw

        blob, path = exchange_unique_id(rank, world, make_id)
        h = C.c_void_p()
        _lib.check(L.nabo_comm_create(C.byref(h), int(device), rank, world, C.c_char_p(blob)))
        return cls(h, int(device), rank, world, path if rank == 0 else None)

    @classmethod
    def all_devices(cls, devices):
        """One process driving several GPUs: a communicator per device (ncclCommInitAll); call their collectives from
        one host thread each (`ShardedGroup` does)."""
        return cls._many(devices, "nabo_comm_create_all")

    @classmethod
    def loopback(cls, devices):
        """N ranks in this process exchanging through device-to-device copies; devices may repeat (N shards on ONE GPU)."""
        return cls._many(devices, "nabo_comm_create_loopback")

    @classmethod
    def _many(cls, devices, fn):
        n = len(devices)
        hs = (C.c_void_p * n)()
        dv = (C.c_int32 * n)(*[int(d) for d in devices])
        _lib.check(getattr(_lib.lib(), fn)(hs, dv, n))
        return [cls(C.c_void_p(hs[i]), int(devices[i]), i, n) for i in range(n)]

    def set_ref_shards(self, ref_shards):
        """2-D layout (include/nabo_knn.h: nabo_comm_set_ref_shards); the caller's index must hold piece rank % ref_shards."""
        _lib.check(_lib.lib().nabo_comm_set_ref_shards(self._h, int(ref_shards)))
        self.ref_shards = int(ref_shards)
        return self

    def set_timeout(self, seconds):
        """How long this rank waits for its peers inside a collective before it aborts the communicator."""
        _lib.check(_lib.lib().nabo_comm_set_timeout(self._h, float(seconds)))
        return self

    def abort(self):
        """Give up on the communicator (any thread): ranks blocked in one of its collectives return an error."""
        if self._h is not None and self._h.value:
            _lib.lib().nabo_comm_abort(self._h)

    def transport_ranks(self):
        """ranks the transport itself reports (ncclCommCount / the loopback rendezvous' size)"""
        n = _lib.lib().nabo_comm_transport_ranks(self._h)
        if n < 0:
            _lib.check(n)
        return int(n)

    def barrier(self):
        _lib.check(_lib.lib().nabo_comm_barrier(self._h))

    def allreduce_max(self, value):
        v = C.c_double(float(value))
        _lib.check(_lib.lib().nabo_comm_allreduce_max_f64(self._h, C.byref(v)))
        return v.value

    def close(self):
        if self._h is not None and self._h.value:
            _lib.lib().nabo_comm_destroy(self._h)
            self._h = C.c_void_p()
            if self._id_path:
                try:
                    os.remove(self._id_path)
                except OSError:
                    pass

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001
            pass


class ShardedIndex:
    """This rank's reference shard (a KnnIndex created with ref_index_base = its first global row) behind a
    communicator: `query_device` is nabo_sharded_query -- collective, device pointers, full [m,k] result on every rank."""

    def __init__(self, comm, index, protocol="auto"):
        self.comm, self.index, self.protocol = comm, index, PROTOCOLS[protocol]

    def query_device(self, x_ptr, m, k, drop_first, out_idx_ptr, out_dist_ptr):
        _lib.check(_lib.lib().nabo_sharded_query(self.comm._h, self.index._h, int(x_ptr), int(m), int(k),
                                                 int(bool(drop_first)), int(out_idx_ptr), int(out_dist_ptr), self.protocol))

    def last_stats(self):
        ms = (C.c_double * 8)()
        cn = (C.c_int64 * 4)()
        _lib.check(_lib.lib().nabo_sharded_last_stats(self.comm._h, ms, cn))
        return {"ms_local": ms[0], "ms_exchange": ms[1], "ms_merge": ms[2], "ms_second": ms[3], "ms_slice": ms[4],
                "ms_gather": ms[5], "ms_total": ms[6], "ms_topk_local": ms[7], "uncertified": int(cn[0]), "candidates": int(cn[1]),
                "protocol": {1: "global", 2: "local"}.get(int(cn[3]), "?")}


_LEAKED = []          # handles of ranks whose thread never came back (ShardedGroup.close)


class ShardedGroup:
    """One process, several ranks: rank i = (device[i], shard i of the references), each driven by its own host thread
    (the C calls release the GIL; the collectives inside nabo_sharded_query rendezvous the threads).
    transport "rccl": one GPU per rank, ncclCommInitAll; "loopback": device-to-device copies, devices may repeat.
    A shard may hold fewer than k + drop_first reference cells (it takes part with absent entries)."""

    def __init__(self, devices, n_ref, g, metric, Y, dist_factor=0.25, ref_mask=None, transport="rccl", protocol="auto",
                 ref_shards=None, timeout=None):
        """ref_shards (optional, divides the number of ranks): the 2-D layout of nabo_comm_set_ref_shards -- the
        references in ref_shards pieces (rank r holds piece r % ref_shards), the target rows in len(devices) / ref_shards
        slices; None = one piece per rank.  timeout (seconds; default NABO_COMM_TIMEOUT_S or 600): how long a rank waits
        for its peers inside a collective, and how long a call of this object waits for its rank threads, before the
        group is torn down with an error instead of hanging."""
        from ._knn import KnnIndex
        self.devices = [int(d) for d in devices]
        N = len(self.devices)
        R = N if ref_shards is None else int(ref_shards)
        if R < 1 or N % R:
            raise ValueError("ref_shards must divide the number of ranks")
        if transport != "loopback" and _lib.device_count() < len(set(self.devices)):
            raise _lib.NaboError("%d ranks over RCCL need %d GPUs, %d visible" % (N, len(set(self.devices)), _lib.device_count()))
        self.timeout = float(timeout if timeout is not None else os.environ.get("NABO_COMM_TIMEOUT_S") or 600.0)
        self.comms = Comm.loopback(self.devices) if transport == "loopback" else Comm.all_devices(self.devices)
        for c in self.comms:
            c.set_ref_shards(R)
            c.set_timeout(self.timeout)
        self.ref_shards = R
        self.indices, self._Y, self._mask = [], [], []
        for r in range(N):
            lo, hi = shard_bounds(n_ref, R, r % R)
            if hi <= lo:
                raise ValueError("ERROR: %d reference cells cannot be cut into %d pieces" % (n_ref, R))
            self.indices.append(KnnIndex(hi - lo, g, metric=metric, dist_factor=dist_factor, ref_index_base=lo,
                                         device=self.devices[r]))
            # the rank's reference rows live in HBM from here on (set_ref borrows them: a step of bench.py re-packs
            # the shard, it does not re-upload it)
            from ._knn import DeviceBuffer
            ys = np.ascontiguousarray(Y[lo:hi], dtype=np.float64)
            self._Y.append(DeviceBuffer(ys.nbytes, self.devices[r]).upload(ys))
            self._mask.append(None if ref_mask is None else np.ascontiguousarray(ref_mask[lo:hi], dtype=np.uint8))
        self.shards = [ShardedIndex(c, ix, protocol) for c, ix in zip(self.comms, self.indices)]
        # one persistent host thread per rank (a step of bench.py must not pay for thread creation)
        self._jobs = [queue.Queue() for _ in range(N)]
        self._done = queue.Queue()
        # (the threads hold the two queues, not `self`: a group that is dropped without close() is still collected, and
        # __del__ then shuts the threads down and frees the GPU memory and the communicators)
        self._threads = [threading.Thread(target=ShardedGroup._worker, args=(self._jobs[r], self._done, r), daemon=True)
                         for r in range(N)]
        for t in self._threads:
            t.start()

    @staticmethod
    def _worker(jobs, done, r):
        while True:
            fn = jobs.get()
            if fn is None:
                return
            try:
                fn(r)
                done.put((r, None))
            except BaseException as e:      # noqa: BLE001
                done.put((r, e))
            del fn                          # (the job closes over the group: do not keep it alive while idle)

    def _each(self, fn):
        """fn(r) on every rank's thread.  If a rank has not come back after `timeout` (+ the C side's own deadline for
        a missing peer) every communicator is aborted, which releases the ranks blocked in a collective; the first
        error is raised."""
        N = len(self.shards)
        for q in self._jobs:
            q.put(fn)
        errs, left = [None] * N, N
        deadline = time.time() + self.timeout + 30.0
        aborted = False
        while left:
            try:
                r, e = self._done.get(timeout=max(0.05, deadline - time.time()))
            except queue.Empty:
                if aborted:
                    raise _lib.NaboError("sharded group: %d rank thread(s) did not return after the communicators were aborted" % left)
                for c in self.comms:
                    c.abort()
                aborted, deadline = True, time.time() + 60.0
                continue
            errs[r], left = e, left - 1
        if aborted:
            raise _lib.NaboError("sharded group: a rank did not return within %.0f s; the communicators were aborted (%s)"
                                 % (self.timeout, next((str(e) for e in errs if e is not None), "no rank reported an error")))
        # the rank that failed on its own carries the message that explains the others' NABO_E_COMM
        own = [e for e in errs if e is not None and "a peer failed" not in str(e)]
        for e in own + [e for e in errs if e is not None]:
            raise e

    def set_ref(self):
        self._each(lambda r: self.indices[r].set_ref(y_device_ptr=self._Y[r].ptr, ref_mask=self._mask[r]))
        return self

    def query_device(self, x_ptrs, m, k, drop_first, out_idx_ptrs, out_dist_ptrs):
        """x_ptrs / out_*_ptrs: one device pointer per rank (on that rank's device), or a single pointer when every
        rank sits on the same device (X is then shared, and every rank writes the same bytes to the same outputs --
        pass per-rank outputs to check that)."""
        N = len(self.shards)
        xs = list(x_ptrs) if isinstance(x_ptrs, (list, tuple)) else [x_ptrs] * N
        oi = list(out_idx_ptrs) if isinstance(out_idx_ptrs, (list, tuple)) else [out_idx_ptrs] * N
        od = list(out_dist_ptrs) if isinstance(out_dist_ptrs, (list, tuple)) else [out_dist_ptrs] * N
        self._each(lambda r: self.shards[r].query_device(xs[r], m, k, drop_first, oi[r], od[r]))

    def query(self, X, k, drop_first=False):
        """host arrays in and out (X replicated to every rank's device; rank 0's copy of the result is returned)"""
        from ._knn import DeviceBuffer
        X = np.ascontiguousarray(X, dtype=np.float64)
        m = X.shape[0]
        bufs = {}
        xs, oi, od = [], [], []
        for d in self.devices:
            if d not in bufs:
                bufs[d] = DeviceBuffer(X.nbytes, d).upload(X)
            xs.append(bufs[d].ptr)
        outs = [(DeviceBuffer(m * k * 8, d), DeviceBuffer(m * k * 8, d)) for d in self.devices]
        try:
            self.query_device(xs, m, k, drop_first, [a.ptr for a, _ in outs], [b.ptr for _, b in outs])
            return outs[0][0].download((m, k), np.int64), outs[0][1].download((m, k), np.float64)
        finally:
            for b in bufs.values():
                b.free()
            for a, b in outs:
                a.free(); b.free()

    def last_stats(self, r=0):
        return self.shards[r].last_stats()

    def close(self):
        """Stop the rank threads, then free every rank's index, reference rows and communicator.  A rank whose thread is
        still inside a HIP / RCCL call (after an aborted or timed-out collective) keeps its handles: destroying them under
        a running call would be a use-after-free -- they are leaked and reported instead."""
        threads = getattr(self, "_threads", [])
        for q in getattr(self, "_jobs", []):
            q.put(None)
        for t in threads:
            t.join(timeout=5.0)
        stuck = [r for r, t in enumerate(threads) if t.is_alive()]
        if stuck:
            for c in getattr(self, "comms", []):
                c.abort()                                      # releases ranks blocked in a collective
            for r in stuck:
                threads[r].join(timeout=30.0)
            stuck = [r for r in stuck if threads[r].is_alive()]
        self._jobs, self._threads = [], []
        for r, ix in enumerate(getattr(self, "indices", [])):
            if r not in stuck:
                ix.close()
        for r, y in enumerate(getattr(self, "_Y", [])):
            if r not in stuck:
                y.free()
        for r, c in enumerate(getattr(self, "comms", [])):
            if r not in stuck:
                c.close()
        if stuck:
            import warnings
            for r in stuck:                                    # keep the objects alive: their __del__ would free them
                _LEAKED.append((self.indices[r], self._Y[r], self.comms[r]))
            warnings.warn("ShardedGroup.close: rank thread(s) %s still inside a device call; their index, reference rows and "
                          "communicator were NOT destroyed (leaked)" % stuck, ResourceWarning)
        self.indices, self._Y, self.comms = [], [], []

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001
            pass


class LoopbackGroup(ShardedGroup):
    """bench.py / tests: N shard-ranks on ONE GPU through the loopback transport."""

    def __init__(self, n_ranks, device, n_ref, g, metric, Y, **kw):
        super().__init__([device] * n_ranks, n_ref, g, metric, Y, transport="loopback", **kw)
