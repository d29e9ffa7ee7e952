"""Python host-side mirror of the C ABI: one `Engine` = one kpilqr_ctx.

Array conventions are those of include/kpilqr.h (the reference's Eigen layout): numpy float64,
C-contiguous, one COLUMN-MAJOR matrix per step, i.e. shapes
    A [B,T,n,n] (A[b,t,c,r] = A(r,c)),  B [B,T,m,n],  l_xx [B,T,n,n],  l_uu [B,T,m,m],
    K [B,T,n,m] (K[b,t,c,r] = K(r,c), K is m x n),  k [B,T,m],  r [B,T+1,nr],  r_x [B,T+1,nr,n] ...
Nothing here computes: every method forwards to libkpilqr.so and raises if it reports an error.
"""
import ctypes as C

import numpy as np

from . import _lib


class KpilqrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"kpilqr error {code}: {msg}")
        self.code = code


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(a.shape)}")
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _DeviceArray:
    """Zero-copy view of a context-owned device buffer (for torch.as_tensor(..., device='cuda'))."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False),
                                         "version": 2, "strides": None}
        self._owner = owner


class _PinnedBlock:
    """One kpilqr_host_alloc allocation.  Every numpy array handed out by Engine.pinned() reaches it through its base
    buffer, so the memory lives as long as ANY view of it does -- also after Engine.close(): a result read from a pinned
    array behind close() is a read of live memory, not of a freed page.  The engine keeps a reference of its own until it
    is closed (the allocation never dies under a transfer the engine still has in flight)."""

    def __init__(self, lib, ptr, nbytes):
        self._lib, self.ptr, self.nbytes = lib, ptr, nbytes

    def __del__(self):
        try:
            if self.ptr:
                self._lib.kpilqr_host_free(None, C.c_void_p(self.ptr))     # ctx = NULL: the context may be gone already
                self.ptr = 0
        except Exception:
            pass


class Engine:
    def __init__(self, dof, m, T, nr, batch=1, n_alpha=6, device=0, stream=None, generic=False, tiled=False, fused=False):
        self._L = _lib.load()
        self.dof, self.n, self.m, self.T, self.nr = dof, 2 * dof, m, T, nr
        self.batch, self.n_alpha, self.device = batch, n_alpha, device
        d = _lib.Dims(dof, m, T, nr, batch, n_alpha, device, (_lib.FLAG_GENERIC_KERNELS if generic else 0) | (_lib.FLAG_TILED_KERNELS if tiled else 0) | (_lib.FLAG_FUSED if fused else 0))
        h = C.c_void_p()
        rc = self._L.kpilqr_create(C.byref(d), C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != 0:
            raise KpilqrError(rc, (self._L.kpilqr_strerror(None) or b"").decode())
        self._h = h
        self._keep = []
        self._pinned = []

    # -- plumbing -------------------------------------------------------------------------------
    def _ck(self, rc):
        if rc < 0:
            raise KpilqrError(rc, (self._L.kpilqr_strerror(self._h) or b"").decode())
        return rc

    def close(self):
        if getattr(self, "_h", None):
            self._L.kpilqr_sync(self._h)
            self._L.kpilqr_destroy(self._h)      # (waits for the chunk streams too: nothing reads the pinned blocks any more)
            self._h = None
            self._pinned = []                    # blocks without outstanding views are freed here, the others with their last view

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self):
        self._ck(self._L.kpilqr_sync(self._h))
        self._keep.clear()

    def resize(self, dof, m, T):
        """kpilqr_resize (iLQR_SVR::Resize): new state / control / horizon sizes in place, allocations re-used."""
        self._ck(self._L.kpilqr_resize(self._h, int(dof), int(m), int(T)))
        self.dof, self.n, self.m, self.T = int(dof), 2 * int(dof), int(m), int(T)
        self._keep.clear()

    @property
    def backward_variant(self):
        return self._L.kpilqr_backward_variant(self._h).decode()

    @property
    def forward_variant(self):
        return self._L.kpilqr_forward_variant(self._h).decode()

    def last_launch(self, which="backward"):
        """kpilqr_last_launch: the form the last backward / forward launch actually took, e.g.
        'mfma_f64_t1_fused:w1:raw:uni:ru0' (waits for the stream)."""
        return self._L.kpilqr_last_launch(self._h, 0 if which == "backward" else 1).decode()

    def device_array(self, which, shape, typestr="<f8"):
        p, sz = C.c_void_p(), C.c_size_t()
        self._ck(self._L.kpilqr_device_ptr(self._h, which, C.byref(p), C.byref(sz)))
        item = int(typestr[2:])
        if int(np.prod(shape)) * item > sz.value:
            raise ValueError("requested view larger than the device buffer")
        return _DeviceArray(p.value, shape, typestr, self)

    # -- STEP 1b --------------------------------------------------------------------------------
    def set_keypoints(self, kp_offsets, kp_times):
        """Per-DoF CSR: kp_times[kp_offsets[b*dof+i]:kp_offsets[b*dof+i+1]] sorted time indices."""
        o = np.ascontiguousarray(kp_offsets, dtype=np.int32)
        t = np.ascontiguousarray(kp_times, dtype=np.int32)
        if o.shape != (self.batch * self.dof + 1,):
            raise ValueError("kp_offsets must have batch*dof+1 entries")
        if t.shape != (int(o[-1]),):
            raise ValueError("kp_times length must equal kp_offsets[-1]")
        self._ck(self._L.kpilqr_set_keypoints(self._h, _ptr(o), _ptr(t if len(t) else np.zeros(1, np.int32))))

    def set_keypoints_rows(self, per_traj):
        """per_traj: list over b of (offs[T+1], cols) -- the reference's keypoints[t] lists."""
        offs, times = rows_to_dof_csr(per_traj, self.dof, self.T)
        self.set_keypoints(offs, times)

    # -- key-point placement on the device ----------------------------------------------
    def upload_states(self, X):
        X = _f64(X, (self.batch, self.T, self.n))
        self._keep.append(X)
        self._ck(self._L.kpilqr_upload_states(self._h, _ptr(X)))

    def generate_keypoints(self, method, min_N, max_N=1, thresholds=None, dt=0.0):
        th = None if thresholds is None else _f64(thresholds, (self.dof,))
        self._ck(self._L.kpilqr_generate_keypoints(self._h, method.encode(), int(min_N), int(max_N), _ptr(th), float(dt)))

    def get_keypoints(self):
        """-> (offsets [batch*dof+1], times) per-DoF CSR currently held by the context."""
        offs = np.zeros(self.batch * self.dof + 1, np.int32)
        total = self._ck(self._L.kpilqr_get_keypoints(self._h, _ptr(offs), None, 0))
        times = np.zeros(max(total, 1), np.int32)
        self._ck(self._L.kpilqr_get_keypoints(self._h, _ptr(offs), _ptr(times), len(times)))
        return offs, times[:total]

    def keypoint_error_test(self, intervals, min_N, threshold):
        """kpilqr_keypoint_error_test: intervals [n][4] = (trajectory, DoF, start, end) -> bool [n] (True: good)."""
        iv = np.ascontiguousarray(intervals, np.int32).reshape(-1, 4)
        good = np.zeros(len(iv), np.uint8)
        self._ck(self._L.kpilqr_keypoint_error_test(self._h, len(iv), _ptr(iv), int(min_N), float(threshold), _ptr(good)))
        return good.astype(bool)

    def upload_fd(self, job_b, job_t, job_col, job_mode, xplus, xminus, job_nom=None, xnom=None, eps=1e-6):
        nj = len(job_t)
        jb = np.ascontiguousarray(job_b, np.int32); jt = np.ascontiguousarray(job_t, np.int32)
        jc = np.ascontiguousarray(job_col, np.int32); jm = np.ascontiguousarray(job_mode, np.uint8)
        xp = _f64(xplus, (nj, self.n)); xm = _f64(xminus, (nj, self.n))
        jn = None if job_nom is None else np.ascontiguousarray(job_nom, np.int32)
        xn = None if xnom is None else _f64(xnom)
        nnom = 0 if xn is None else xn.shape[0]
        self._keep += [jb, jt, jc, jm, xp, xm, jn, xn]
        self._ck(self._L.kpilqr_upload_fd(self._h, nj, _ptr(jb), _ptr(jt), _ptr(jc), _ptr(jm), _ptr(jn),
                                          _ptr(xp), _ptr(xm), nnom, _ptr(xn), float(eps)))

    # -- asynchronous boundary: pinned memory, one slab, chunk pipeline -------------------------------
    def pinned(self, shape, dtype=np.float64):
        """numpy array in pinned host memory (kpilqr_host_alloc).  The allocation is released when the engine has been closed
        AND no array (or view of one) refers to it any more -- whichever comes last."""
        shape = (shape,) if np.isscalar(shape) else tuple(shape)
        nbytes = max(int(np.prod(shape)) * np.dtype(dtype).itemsize, 1)
        p = C.c_void_p()
        self._ck(self._L.kpilqr_host_alloc(self._h, nbytes, C.byref(p)))
        block = _PinnedBlock(self._L, p.value, nbytes)
        self._pinned.append(block)
        buf = (C.c_char * nbytes).from_address(p.value)
        buf._kpilqr_block = block                # numpy keeps `buf` as the base of every view; `buf` keeps the block
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def fd_slab(self, job_b, job_t, job_col, job_mode, xplus, xminus, job_nom=None, xnom=None):
        """Packs FD jobs (sorted by trajectory for the chunk pipeline) into ONE pinned slab in the layout of
        kpilqr_fd_slab_layout.  Returns a dict for upload_fd_slab / iterate_streamed."""
        jb = np.asarray(job_b, np.int32); jt = np.asarray(job_t, np.int32)
        nj = len(jt)
        nnom = 0 if xnom is None else len(xnom)
        lay = _lib.FdLayout()
        self._ck(self._L.kpilqr_fd_slab_layout(self._h, nj, nnom, C.byref(lay)))
        slab = self.pinned(lay.bytes, np.uint8)

        def put(off, a, dt):
            a = np.ascontiguousarray(a, dt)
            slab[off:off + a.nbytes] = a.view(np.uint8).reshape(-1)
        put(lay.xplus, xplus, np.float64); put(lay.xminus, xminus, np.float64)
        if nnom:
            put(lay.xnom, xnom, np.float64)
        put(lay.job_b, jb, np.int32); put(lay.job_t, jt, np.int32); put(lay.job_col, job_col, np.int32)
        put(lay.job_nom, np.zeros(nj, np.int32) if job_nom is None else job_nom, np.int32)
        put(lay.job_mode, job_mode, np.uint8)
        # per-trajectory offsets for the chunk pipeline
        tjf = self.pinned(self.batch + 1, np.int32); tjf[:] = np.searchsorted(jb, np.arange(self.batch + 1))
        tnf = self.pinned(self.batch + 1, np.int32)
        if nnom and job_nom is not None:
            jn = np.asarray(job_nom, np.int64)
            lo = np.full(self.batch + 1, nnom, np.int64)
            np.minimum.at(lo, jb, jn)
            for b in range(self.batch - 1, -1, -1):
                lo[b] = min(lo[b], lo[b + 1])
            lo[0] = 0
            tnf[:] = lo
        else:
            tnf[:] = 0
        return dict(slab=slab, njobs=nj, nnom=nnom, traj_job_first=tjf, traj_nom_first=tnf, layout=lay)

    def fd_kp_slab(self, xplus, xminus, mode, pinned=True):
        """Key-point ordered payload (kpilqr_fd_kp_layout): xplus / xminus [entries][3][n], mode [entries] (bit k: kind k is
        one-sided), packed into ONE slab -- pinned (the asynchronous boundary) or, for a one-off upload of a large resident
        payload, pageable.  Returns a dict for upload_fd_kp / iterate_streamed(fd_kp=...)."""
        ent = int(np.shape(mode)[0])
        lay = _lib.FdkpLayout()
        self._ck(self._L.kpilqr_fd_kp_layout(self._h, ent, C.byref(lay)))
        slab = self.pinned(lay.bytes, np.uint8) if pinned else np.zeros(max(lay.bytes, 1), np.uint8)
        n3 = 3 * self.n
        rec = slab[:ent * lay.entry_stride].view(np.float64).reshape(ent, lay.entry_stride // 8)     # one record per entry
        es = max(1, lay.elem_stride // 8)                # doubles from one element of a side to the next (2: x+ and x- side by side)
        for off, a in ((lay.xplus, xplus), (lay.xminus, xminus)):
            a = np.ascontiguousarray(a, np.float64)
            if a.shape != (ent, 3, self.n):
                raise ValueError(f"expected shape {(ent, 3, self.n)}, got {a.shape}")
            rec[:, off // 8:off // 8 + n3 * es:es] = a.reshape(ent, n3)
        rec[:, lay.mode // 8:] = 0.0
        rec[:, lay.mode // 8:lay.mode // 8 + 1].view(np.int32)[:, 0] = np.asarray(mode, np.int32)
        return dict(slab=slab, entries=ent, layout=lay)

    def upload_fd_kp(self, s, eps=1e-6):
        self._ck(self._L.kpilqr_upload_fd_kp(self._h, _ptr(s["slab"]), s["entries"], float(eps)))

    def kp_columns(self, xplus, xminus, mode, eps=1e-6, pinned=True):
        """The key-point columns differenced ON THE HOST from the arrays of fd_kp_slab (IEEE quotients: bit for bit what
        the device forms from the same payload): [entries][3][n] for upload_kp_columns / iterate_streamed(kp_cols=...)."""
        ent = int(np.shape(mode)[0])
        xp = np.ascontiguousarray(xplus, np.float64).reshape(ent, 3, self.n)
        xm = np.ascontiguousarray(xminus, np.float64).reshape(ent, 3, self.n)
        den = np.where((np.asarray(mode, np.int32)[:, None] >> np.arange(3)[None, :]) & 1, float(eps), 2.0 * float(eps))   # [entries][kind]
        cols = self.pinned(max(ent * 3 * self.n, 1)) if pinned else np.zeros(max(ent * 3 * self.n, 1))
        cols[:ent * 3 * self.n].reshape(ent, 3, self.n)[...] = (xp - xm) / den[:, :, None]
        return dict(cols=cols, entries=ent)

    def upload_kp_columns(self, s):
        self._ck(self._L.kpilqr_upload_kp_columns(self._h, _ptr(s["cols"]), s["entries"]))

    def upload_fd_slab(self, s, eps=1e-6):
        self._ck(self._L.kpilqr_upload_fd_slab(self._h, _ptr(s["slab"]), s["njobs"], s["nnom"], float(eps)))

    def iterate_streamed(self, fd=None, fd_kp=None, kp_cols=None, eps=1e-6, r=None, r_x=None, r_u=None, u_nom=None, lam=None, K=None, k=None,
                         cost_pred=None, delta_J=None, status=None, pd_stride=100, nchunks=0):
        """kpilqr_iterate_streamed: every array must come from self.pinned(); asynchronous (sync() to wait)."""
        io = _lib.StreamIO()
        if fd is not None:
            io.fd_slab = fd["slab"].ctypes.data; io.njobs = fd["njobs"]; io.nnom = fd["nnom"]
            io.traj_job_first = fd["traj_job_first"].ctypes.data; io.traj_nom_first = fd["traj_nom_first"].ctypes.data
        if fd_kp is not None:
            io.fd_kp_slab = fd_kp["slab"].ctypes.data; io.entries = fd_kp["entries"]
        if kp_cols is not None:
            io.kp_columns = kp_cols["cols"].ctypes.data; io.entries = kp_cols["entries"]
        io.eps = float(eps)
        for name, a in (("r", r), ("r_x", r_x), ("r_u", r_u), ("u_nom", u_nom), ("lam", lam), ("K", K), ("k", k),
                        ("cost_pred", cost_pred), ("delta_J", delta_J), ("status", status)):
            if a is not None:
                setattr(io, name, a.ctypes.data)
        self._ck(self._L.kpilqr_iterate_streamed(self._h, C.byref(io), int(pd_stride), int(nchunks)))

    def fd_difference(self):
        self._ck(self._L.kpilqr_fd_difference(self._h))

    def interpolate(self):
        self._ck(self._L.kpilqr_interpolate(self._h))

    def filter_dynamics(self, method, coefs):
        """Optimiser::FilterDynamicsMatrices on the materialised A (after interpolate)."""
        co = _f64(coefs)
        self._ck(self._L.kpilqr_filter_dynamics(self._h, method.encode(), _ptr(co), len(co)))

    def dof_importance(self, sampling_k_interval=1):
        """iLQR_SVR::LeastImportantDofs (summing branch) over the last gains -> [batch][dof]."""
        out = np.zeros((self.batch, self.dof))
        self._ck(self._L.kpilqr_dof_importance(self._h, int(sampling_k_interval), _ptr(out)))
        self.sync()
        return out

    # -- STEP 1c --------------------------------------------------------------------------------
    def upload_residuals(self, r=None, r_x=None, r_u=None, w_run=None, w_term=None):
        B, T1, n, m, nr = self.batch, self.T + 1, self.n, self.m, self.nr
        r = None if r is None else _f64(r, (B, T1, nr))
        r_x = None if r_x is None else _f64(r_x, (B, T1, nr, n))
        r_u = None if r_u is None else _f64(r_u, (B, T1, nr, m))
        w_run = None if w_run is None else _f64(w_run, (nr,))
        w_term = None if w_term is None else _f64(w_term, (nr,))
        self._keep += [r, r_x, r_u, w_run, w_term]
        self._ck(self._L.kpilqr_upload_residuals(self._h, _ptr(r), _ptr(r_x), _ptr(r_u), _ptr(w_run), _ptr(w_term)))

    def upload_residual_jacobians_const(self, r_x, r_u=None):
        """kpilqr_upload_residual_jacobians_const: ONE r_x [nr][n] (and r_u [nr][m], or None for r_u = 0) for every trajectory
        and step."""
        r_x = _f64(r_x, (self.nr, self.n))
        r_u = None if r_u is None else _f64(r_u, (self.nr, self.m))
        self._keep += [r_x, r_u]
        self._ck(self._L.kpilqr_upload_residual_jacobians_const(self._h, _ptr(r_x), _ptr(r_u)))

    def cost_derivs(self):
        self._ck(self._L.kpilqr_cost_derivs(self._h))

    def trajectory_cost(self):
        out = np.zeros(self.batch)
        self._ck(self._L.kpilqr_trajectory_cost(self._h, _ptr(out)))
        self.sync()
        return out

    # -- STEP 2 ---------------------------------------------------------------------------------
    def backward(self, lam, pd_stride=100, fetch=True):
        if lam is not None:         # None: keep the lambdas already resident on the device
            lam = _f64(np.broadcast_to(np.asarray(lam, dtype=np.float64), (self.batch,)))
            self._keep.append(lam)
        if not fetch:
            self._ck(self._L.kpilqr_backward(self._h, _ptr(lam), pd_stride, None, None))
            return None
        status = np.zeros(self.batch, np.int32); dJ = np.zeros(self.batch)
        self._ck(self._L.kpilqr_backward(self._h, _ptr(lam), pd_stride, _ptr(status), _ptr(dJ)))
        self.sync()
        return status, dJ

    def backward_stats(self, pd_stride=100):
        """kpilqr_backward_stats: the instrumented backward sweep of a fused context -> hist [batch][6] (see include/kpilqr.h)."""
        h = np.zeros((self.batch, 6), np.int32)
        self._ck(self._L.kpilqr_backward_stats(self._h, int(pd_stride), _ptr(h)))
        return h

    def gains(self):
        K = np.zeros((self.batch, self.T, self.n, self.m)); k = np.zeros((self.batch, self.T, self.m))
        self._ck(self._L.kpilqr_download_gains(self._h, _ptr(K), _ptr(k)))
        self.sync()
        return K, k

    # -- STEP 3 ---------------------------------------------------------------------------------
    def upload_nominal(self, u_nom=None, ctrl_lim=None):
        u = None if u_nom is None else _f64(u_nom, (self.batch, self.T, self.m))
        cl = None if ctrl_lim is None else _f64(ctrl_lim, (2 * self.m,))
        self._keep += [u, cl]
        self._ck(self._L.kpilqr_upload_nominal(self._h, _ptr(u), _ptr(cl)))

    def forward_linear(self, alphas, want_U=False, fetch=True):
        a = None if alphas is None else _f64(alphas, (self.n_alpha,))   # None: keep resident alphas
        self._keep.append(a)
        if not fetch:
            self._ck(self._L.kpilqr_forward_linear(self._h, _ptr(a), None, None))
            return None
        cost = np.zeros((self.batch, self.n_alpha))
        U = np.zeros((self.batch, self.n_alpha, self.T, self.m)) if want_U else None
        self._ck(self._L.kpilqr_forward_linear(self._h, _ptr(a), _ptr(cost), _ptr(U)))
        self.sync()
        return (cost, U) if want_U else cost

    # -- several GPUs: the line-search cost reduction through the C ABI (RCCL) ---------------------------
    def comm_unique_id(self):
        buf = (C.c_char * 128)()
        self._ck(self._L.kpilqr_comm_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    def comm_init(self, nranks, rank, unique_id):
        buf = C.create_string_buffer(unique_id, 128)
        self._ck(self._L.kpilqr_comm_init(self._h, int(nranks), int(rank), C.cast(buf, C.c_void_p)))

    def allreduce_linesearch(self):
        out = np.zeros(8)
        self._ck(self._L.kpilqr_allreduce_linesearch(self._h, _ptr(out)))
        self.sync()
        return out

    def iterate(self, lam=None, pd_stride=100, alphas=None):
        """Enqueue STEP 1b + 1c + 2 + 3 back to back (asynchronous)."""
        lam = None if lam is None else _f64(np.broadcast_to(np.asarray(lam, dtype=np.float64), (self.batch,)))
        a = None if alphas is None else _f64(alphas, (self.n_alpha,))
        self._keep += [lam, a]
        self._ck(self._L.kpilqr_iterate(self._h, _ptr(lam), pd_stride, _ptr(a)))

    def results(self):
        """Device-resident outputs of the last backward/forward: status, delta_J, cost_pred."""
        import ctypes
        out = {}
        self.sync()
        for name, which, shape, dt in (("status", _lib.BUF_STATUS, (self.batch,), np.int32),
                                       ("delta_J", _lib.BUF_DELTA_J, (self.batch,), np.float64),
                                       ("cost_pred", _lib.BUF_COST_PRED, (self.batch, self.n_alpha), np.float64)):
            out[name] = self._d2h(which, shape, dt)
        return out

    def _d2h(self, which, shape, dt):
        import torch
        t = torch.as_tensor(self.device_array(which, shape, "<i4" if dt == np.int32 else "<f8"),
                            device=f"cuda:{self.device}")
        return t.cpu().numpy().copy()

    # -- debug / oracle hooks ---------------------------------------------------------------------
    def set_AB(self, A=None, B=None):
        Bt, T, n, m = self.batch, self.T, self.n, self.m
        A = None if A is None else _f64(A, (Bt, T, n, n)); B = None if B is None else _f64(B, (Bt, T, m, n))
        self._ck(self._L.kpilqr_set_AB(self._h, _ptr(A), _ptr(B)))

    def get_AB(self):
        Bt, T, n, m = self.batch, self.T, self.n, self.m
        A = np.zeros((Bt, T, n, n)); B = np.zeros((Bt, T, m, n))
        self._ck(self._L.kpilqr_get_AB(self._h, _ptr(A), _ptr(B)))
        return A, B

    def set_cost_derivs(self, l_x=None, l_xx=None, l_u=None, l_uu=None):
        Bt, T, n, m = self.batch, self.T, self.n, self.m
        l_x = None if l_x is None else _f64(l_x, (Bt, T, n)); l_xx = None if l_xx is None else _f64(l_xx, (Bt, T, n, n))
        l_u = None if l_u is None else _f64(l_u, (Bt, T, m)); l_uu = None if l_uu is None else _f64(l_uu, (Bt, T, m, m))
        self._ck(self._L.kpilqr_set_cost_derivs(self._h, _ptr(l_x), _ptr(l_xx), _ptr(l_u), _ptr(l_uu)))

    def get_cost_derivs(self):
        Bt, T, n, m = self.batch, self.T, self.n, self.m
        l_x = np.zeros((Bt, T, n)); l_xx = np.zeros((Bt, T, n, n)); l_u = np.zeros((Bt, T, m)); l_uu = np.zeros((Bt, T, m, m))
        self._ck(self._L.kpilqr_get_cost_derivs(self._h, _ptr(l_x), _ptr(l_xx), _ptr(l_u), _ptr(l_uu)))
        return l_x, l_xx, l_u, l_uu


def rows_to_dof_csr(per_traj, dof, T):
    """Reference key-point rows (CSR over time, one (offs, cols) pair per trajectory) -> the C ABI's
    per-DoF CSR of sorted, de-duplicated time indices (offsets [B*dof+1], times)."""
    offs = [0]
    times = []
    for (o, c) in per_traj:
        o = np.asarray(o); c = np.asarray(c)
        t_of = np.repeat(np.arange(T), np.diff(o))
        for i in range(dof):
            ti = np.unique(t_of[c == i])
            times.append(ti.astype(np.int32))
            offs.append(offs[-1] + len(ti))
    return np.asarray(offs, np.int32), (np.concatenate(times) if times else np.zeros(0, np.int32))
