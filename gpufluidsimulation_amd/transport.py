"""Communicator set-up for z-slab ranks (one process per GPU).

Two transports behind the same C-ABI (include/bimocq_gpu.h, section 4):
  * RCCL (default on a multi-GPU node): rank 0 asks the library for an ncclUniqueId, the id travels
    through the torch.distributed store, every rank calls fl_comm_init -- after that ghost planes move
    GPU-to-GPU over xGMI with ncclSend/ncclRecv issued by the C++ solver itself;
  * host-staged over torch.distributed (gloo): fl_comm_set_custom with the two callbacks below.  Slow,
    but runs anywhere -- several ranks on one GPU, or the CPU stand-in of the ABI in the tests.
"""
import ctypes as C

import numpy as np

EXCHANGE_CB = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int),
                          C.c_int, C.c_int, C.c_int)
ALLREDUCE_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int)
P2P_CB = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                     C.POINTER(C.c_void_p), C.POINTER(C.c_size_t))


def init_rccl(lib, dist):
    """RCCL transport: distribute the unique id through the process group's store."""
    rank, world = dist.get_rank(), dist.get_world_size()
    buf = (C.c_ubyte * 128)()
    if rank == 0:
        rc = lib.fl_comm_unique_id(buf)
        if rc != 0:
            raise RuntimeError(f"fl_comm_unique_id failed: {lib.fl_last_error_string()}")
    payload = [bytes(buf)]
    dist.broadcast_object_list(payload, src=0)
    ident = (C.c_ubyte * 128).from_buffer_copy(payload[0])
    rc = lib.fl_comm_init(ident, rank, world)
    if rc != 0:
        raise RuntimeError(f"fl_comm_init failed: {lib.fl_last_error_string()}")


class HostStagedTransport:
    """Ghost-plane exchange and scalar all-reduce over a torch.distributed group, staged through host
    memory with the library's own fl_memcpy_d2h / fl_memcpy_h2d.  Plane ranges follow
    fl_halo_exchange (csrc/bq_halo.hip): with own = nk_local - 2G,
        to   rank-1: local planes [G, G+depth+extra)          from rank-1: into [G-depth, G)
        to   rank+1: local planes [G+own-depth, G+own)        from rank+1: into [G+own, G+own+depth+extra)
    """

    def __init__(self, lib, dist, group=None):
        import torch
        self.torch, self.lib, self.dist, self.group = torch, lib, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._ex = EXCHANGE_CB(self._exchange)
        self._ar = ALLREDUCE_CB(self._allreduce)
        self._pp = P2P_CB(self._p2p)
        lib.fl_comm_set_custom.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.fl_comm_set_custom(self.rank, self.world, C.cast(self._ex, C.c_void_p), C.cast(self._ar, C.c_void_p))
        lib.fl_comm_set_custom_p2p.argtypes = [C.c_void_p]
        lib.fl_comm_set_custom_p2p(C.cast(self._pp, C.c_void_p))
        self.exchanges = 0
        self.planes_moved = 0
        self.p2p_messages = 0
        self.p2p_floats = 0
        self.trace = None                       # set to [] to record (fields, depth, bytes sent per neighbour) per exchange

    def _get(self, ptr, offset_elems, count):
        out = np.empty(count, dtype=np.float32)
        self.lib.fl_memcpy_d2h(C.c_void_p(out.ctypes.data), C.c_void_p(ptr + 4 * offset_elems), C.c_size_t(4 * count))
        return out

    def _put(self, ptr, offset_elems, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        self.lib.fl_memcpy_h2d(C.c_void_p(ptr + 4 * offset_elems), C.c_void_p(arr.ctypes.data), C.c_size_t(4 * arr.size))

    def _exchange(self, n, fields, plane_elems, extra, nk_local, G, depth):
        torch, dist = self.torch, self.dist
        own = nk_local - 2 * G
        lo, hi = self.rank - 1, self.rank + 1
        reqs, recvs, keep = [], [], []          # keep: send buffers must outlive the isend
        for f in range(n):
            ptr, pe, ex = fields[f], plane_elems[f], extra[f]
            if lo >= 0:
                send = torch.from_numpy(self._get(ptr, pe * G, pe * (depth + ex)))
                keep.append(send)
                recv = torch.empty(pe * depth, dtype=torch.float32)
                reqs.append(dist.isend(send, lo, group=self.group, tag=2 * f))
                reqs.append(dist.irecv(recv, lo, group=self.group, tag=2 * f + 1))
                recvs.append((ptr, pe * (G - depth), recv))
            if hi < self.world:
                send = torch.from_numpy(self._get(ptr, pe * (G + own - depth), pe * depth))
                keep.append(send)
                recv = torch.empty(pe * (depth + ex), dtype=torch.float32)
                reqs.append(dist.isend(send, hi, group=self.group, tag=2 * f + 1))
                reqs.append(dist.irecv(recv, hi, group=self.group, tag=2 * f))
                recvs.append((ptr, pe * (G + own), recv))
        for r in reqs:
            r.wait()
        for ptr, off, recv in recvs:
            self._put(ptr, off, recv.numpy())
        self.exchanges += 1
        self.planes_moved += n * depth
        if self.trace is not None:
            self.trace.append((n, depth, 4 * sum(plane_elems[f] * depth for f in range(n))))

    def _p2p(self, n, peers, send, send_count, recv, recv_count):
        """fl_p2p_exchange: message m goes to / comes from rank peers[m] (wall sheets; any pair of ranks)"""
        torch, dist = self.torch, self.dist
        reqs, recvs, keep = [], [], []
        counts = {}

        def seq(peer, sending):             # k-th message to / from this peer inside one exchange: its own tag
            k = counts.get((peer, sending), 0)
            counts[(peer, sending)] = k + 1
            return k
        self._seq = seq
        for m in range(n):
            peer, ns, nr = peers[m], send_count[m], recv_count[m]
            if ns:
                t = torch.from_numpy(self._get(send[m], 0, ns))
                keep.append(t)
                reqs.append(dist.isend(t, peer, group=self.group, tag=1000 + self._seq(peer, True)))
            if nr:
                r = torch.empty(nr, dtype=torch.float32)
                reqs.append(dist.irecv(r, peer, group=self.group, tag=1000 + self._seq(peer, False)))
                recvs.append((recv[m], r))
        for r in reqs:
            r.wait()
        for ptr, r in recvs:
            self._put(ptr, 0, r.numpy())
            self.p2p_floats += r.numel()
        self.p2p_messages += n

    def _allreduce(self, host, count, is_double, is_max):
        torch, dist = self.torch, self.dist
        ctype = C.c_double if is_double else C.c_float
        view = np.ctypeslib.as_array(C.cast(host, C.POINTER(ctype)), shape=(count,))
        t = torch.from_numpy(view.copy())
        dist.all_reduce(t, op=dist.ReduceOp.MAX if is_max else dist.ReduceOp.SUM, group=self.group)
        view[:] = t.numpy()


class NullTransport:
    """Timing aid, not a transport (fl_comm_set_null): ONE process runs the z-slab code path of rank `rank` of `nranks`
    (ghost planes, chunked Jacobi, split operators) with exchanges that move nothing and never wait, so that its
    compute-side cost can be measured on a single GPU.  The fields it produces are meaningless near the slab boundary."""

    def __init__(self, lib, rank=0, nranks=2):
        lib.fl_comm_set_null(rank, nranks)
