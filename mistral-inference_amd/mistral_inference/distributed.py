"""Pipeline-parallel transport: RCCL over xGMI through the C ABI (`mi_rccl_*`, include/mistral_hip.h).

The reference moves activations between pipeline stages and broadcasts the logits with `torch.distributed`
(transformer.py:196,214,237).  `RcclComm` issues the same three exchanges as stream-ordered ncclSend / ncclRecv /
ncclBroadcast on the CURRENT HIP stream: nothing passes through Python's process-group layer per token, and the calls
can be captured in the decode step's hipGraph.  `torch.distributed` remains the bootstrap channel (it carries the
128-byte rendezvous id once) and the transport of CPU tests (gloo).
"""
from __future__ import annotations

import ctypes as C
import logging
import os
from typing import Optional

import torch

from . import _hip


class RcclComm:
    """One RCCL communicator for this process (one process per GPU)."""

    def __init__(self, world_size: int, rank: int, unique_id: bytes):
        assert len(unique_id) == 128
        self.world_size, self.rank = world_size, rank
        self._h = C.c_void_p()
        buf = C.create_string_buffer(unique_id, 128)
        _hip.check(_hip.lib().mi_rccl_init(C.byref(self._h), world_size, rank, buf), "mi_rccl_init")

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _hip.check(_hip.lib().mi_rccl_unique_id(buf), "mi_rccl_unique_id")
        return buf.raw

    @classmethod
    def from_process_group(cls) -> "RcclComm":
        """Rendezvous through the already-initialised torch.distributed group (reference main.py:110-118): rank 0's id
        travels as ONE object broadcast, after which the group is no longer on the hot path."""
        dist = torch.distributed
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(world, rank, box[0])

    # -- the three exchanges of the pipeline (tensors: contiguous, on this process's device) ------------------------
    def send(self, t: torch.Tensor, dst: int) -> None:
        assert t.is_cuda and t.is_contiguous()
        _hip.check(_hip.lib().mi_rccl_send(self._h, t.data_ptr(), t.numel() * t.element_size(), dst, _hip.stream_ptr(t.device)),
                   "mi_rccl_send")

    def recv(self, t: torch.Tensor, src: int) -> None:
        assert t.is_cuda and t.is_contiguous()
        _hip.check(_hip.lib().mi_rccl_recv(self._h, t.data_ptr(), t.numel() * t.element_size(), src, _hip.stream_ptr(t.device)),
                   "mi_rccl_recv")

    def broadcast(self, t: torch.Tensor, src: int) -> None:
        assert t.is_cuda and t.is_contiguous()
        _hip.check(_hip.lib().mi_rccl_bcast(self._h, t.data_ptr(), t.numel() * t.element_size(), src, _hip.stream_ptr(t.device)),
                   "mi_rccl_bcast")

    def exchange(self, send_t: Optional[torch.Tensor], dst: int, recv_t: Optional[torch.Tensor], src: int) -> None:
        """One grouped send + recv (ncclGroupStart / End): the two progress together, so a ring of ranks that all send
        forward and receive from behind cannot deadlock on rendezvous (pipeline_decode.InterleavedDecoder)."""
        L = _hip.lib()
        _hip.check(L.mi_rccl_group_start(), "mi_rccl_group_start")
        if send_t is not None:
            self.send(send_t, dst)
        if recv_t is not None:
            self.recv(recv_t, src)
        _hip.check(L.mi_rccl_group_end(), "mi_rccl_group_end")

    def exchange_with_self(self, src: torch.Tensor, dst: torch.Tensor) -> None:
        """Grouped send + recv to this very rank (the single-GPU test of the transport)."""
        L = _hip.lib()
        _hip.check(L.mi_rccl_group_start(), "mi_rccl_group_start")
        self.send(src, self.rank)
        self.recv(dst, self.rank)
        _hip.check(L.mi_rccl_group_end(), "mi_rccl_group_end")

    def close(self) -> None:
        if self._h:
            _hip.check(_hip.lib().mi_rccl_destroy(self._h), "mi_rccl_destroy")
            self._h = C.c_void_p()


class TorchDistComm:
    """The reference's own transport (torch.distributed send / recv / broadcast): CPU tests over gloo, and the fallback
    selected with MI_PP_TRANSPORT=torch."""

    @staticmethod
    def _order_behind_stream(t: torch.Tensor) -> None:
        """Test rigs only (ranks sharing one GPU over gloo): gloo reads and writes a device tensor from the host without ordering
        itself behind the stream that produces / still reads it, so a hop could pick up a hidden state its kernel had not finished
        (round 6: run-to-run different log-probabilities in tests/test_gpu_pipeline.py, one 0.08 outlier).  The stream is drained
        first; under NCCL (= RCCL) the transfer is stream-ordered and nothing is done here."""
        if t.is_cuda and torch.distributed.get_backend() != "nccl":
            torch.cuda.current_stream(t.device).synchronize()

    def send(self, t: torch.Tensor, dst: int) -> None:
        self._order_behind_stream(t)
        torch.distributed.send(t, dst=dst)

    def recv(self, t: torch.Tensor, src: int) -> None:
        self._order_behind_stream(t)
        torch.distributed.recv(t, src=src)

    def broadcast(self, t: torch.Tensor, src: int) -> None:
        self._order_behind_stream(t)
        torch.distributed.broadcast(t, src=src)

    def exchange(self, send_t: Optional[torch.Tensor], dst: int, recv_t: Optional[torch.Tensor], src: int) -> None:
        """Grouped send + recv (`batch_isend_irecv`: one NCCL group on a GPU, two non-blocking requests over gloo)."""
        dist = torch.distributed
        t = send_t if send_t is not None else recv_t
        if t is not None:
            self._order_behind_stream(t)
        ops = []
        if send_t is not None:
            ops.append(dist.P2POp(dist.isend, send_t, dst))
        if recv_t is not None:
            ops.append(dist.P2POp(dist.irecv, recv_t, src))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()  # (NCCL: orders the current stream behind the transfer; gloo: blocks until the data is there)


def pipeline_comm(device: torch.device):
    """Transport for a pipeline rank on `device`: RCCL through the C ABI on a GPU under the "nccl" process group (the
    opt-in path, MI_PP_TRANSPORT=rccl), torch.distributed otherwise (default).  Whether librccl can be loaded is probed locally on every rank
    (`mi_rccl_unique_id` needs no peer) and AGREED through the process group before anyone enters the collective
    communicator init: a rank that cannot load it would otherwise leave the others waiting in `ncclCommInitRank`."""
    # Default: torch.distributed ("nccl" there IS RCCL over xGMI).  The C-ABI communicator below is a second RCCL
    # communicator that has only ever run at world size 1 (tests/test_gpu_rccl.py - a 1-GPU lease cannot host two ranks);
    # until a multi-GPU run has covered send/recv/bcast and graph replay it is opt-in: MI_PP_TRANSPORT=rccl.  Since a
    # stage's decode step is ONE launch on the persistent engine, the eager torch hop costs a few microseconds per token.
    want = os.environ.get("MI_PP_TRANSPORT", "torch")
    dist = torch.distributed
    if want == "rccl" and device.type == "cuda" and dist.is_initialized() and dist.get_backend() == "nccl":
        err = None
        try:
            RcclComm.unique_id()
        except (RuntimeError, OSError, AttributeError) as e:
            err = e
        flag = torch.tensor([0 if err else 1], device=device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            return RcclComm.from_process_group()
        logging.getLogger(__name__).warning(
            "RCCL through the C ABI is unavailable on at least one rank (%s): pipeline hops use torch.distributed", err)
    return TorchDistComm()
