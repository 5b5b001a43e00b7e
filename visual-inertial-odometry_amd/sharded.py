"""Landmark-sharded solve over the GPUs of one node.

One process per GPU.  Every rank holds a block of the landmarks with all their observations; poses,
speed-biases, extrinsic, pre-integrations and prior are replicated.  Per linearisation ONE exchange of the packed
partial reduced visual system (the 78 upper blocks of the 72x72 H, reduced b, direct b, direct diagonal, chi2, the step's
gain-ratio partial, max |h_ll|: 3028 fp64 = 24 KB per rank; the library says how many, vio_exchange_buffers) and, on the
stepwise path only, per trial step one of two scalars (chi2 of the trial state, landmark part of the gain-ratio
denominator); every rank then runs the identical damped LDLT and updates its own landmarks.

The exchange is an ALL-GATHER into rank-major receive buffers (vio_gather_buffers), and the library adds the ranks' slabs in
rank order wherever it reads a sum: the same additions in the same order on every rank, so all ranks hold bit-identical
systems and take bit-identical LM decisions by construction — not because the collective library happens to pick an
algorithm that returns the same bits everywhere.  The collective is RCCL over xGMI on the GPU box: by default the library
calls ncclAllGather itself on its own stream (vio_comm_init; the communicator id is broadcast once through
torch.distributed).  The portable alternative — and the gloo path of the CPU tests — is vio_set_exchange_hook: the library
calls back at the points of the LM loop where the exchange belongs and the hook runs torch.distributed.all_gather on the
bound buffers.
"""
import numpy as np

from . import synth


class ShardedBackend:
    def __init__(self, lib, window, rank, world, dist=None, torch_device="cuda", ctx_kwargs=None, force_hook=False,
                 exchange=None):
        """exchange: "native" — the library all-gathers with RCCL itself, in stream order, no Python in the loop
        (HIP library on GPUs; the 128-byte communicator id travels through torch.distributed once);
        "hook" — torch.distributed.all_gather_into_tensor from the library's exchange hook (any backend: gloo in the CPU tests);
        "hook_host" — the same hook staging the buffers through pinned host memory: gloo between processes whose tensors
        live on a GPU (RCCL refuses two ranks on one device; this is how the two-rank protocol runs on a one-GPU box).
        Default: native when the library exports it and the device is a GPU, unless VIO_EXCHANGE says otherwise."""
        import torch
        self.torch = torch
        self.dist = dist
        self.rank, self.world = rank, world
        self.full = window
        self.shard = synth.shard_window(window, rank, world) if world > 1 else window
        kw = dict(ctx_kwargs or {})
        kw.update(shard_rank=rank, shard_count=world)
        if exchange is None:
            import os
            exchange = os.environ.get("VIO_EXCHANGE", "native" if (lib.has("comm_init") and str(torch_device).startswith("cuda")) else "hook")
        self.stream = None
        if exchange in ("hook", "hook_host") and str(torch_device).startswith("cuda") and not kw.get("stream"):
            # the hook's collective / copies must be ordered with the library's kernels: one explicit torch stream for
            # both (torch's default stream is the null stream, whose handle 0 the library reads as "create your own")
            self.stream = torch.cuda.Stream()
            kw["stream"] = self.stream.cuda_stream
        self.ctx = lib.context(**kw)
        self.ctx.load(self.shard)
        (_, self.n_red), (_, self.n_sc) = self.ctx.exchange_buffers()
        # caller-owned exchange buffers so that the collective runs in place on them: send side ...
        self.red = torch.zeros(self.n_red + 8, dtype=torch.float64, device=torch_device)
        self.sca = torch.zeros(8, dtype=torch.float64, device=torch_device)
        self.ctx.bind_exchange_buffers(self.red.data_ptr(), self.sca.data_ptr())
        # ... and receive side, rank-major: [world][n_red], [world][n_sc]
        self.gred = torch.zeros(world * self.n_red, dtype=torch.float64, device=torch_device)
        self.gsca = torch.zeros(world * self.n_sc, dtype=torch.float64, device=torch_device)
        self.ctx.bind_gather_buffers(self.gred.data_ptr(), self.gsca.data_ptr())
        self._send = (self.red[:self.n_red], self.sca[:self.n_sc])
        self._recv = (self.gred, self.gsca)
        self._max_view = self.sca[2:3]
        self.exchange = exchange if (world > 1 or force_hook) else "none"
        if self.exchange == "native":
            # ncclCommInitRank is collective: a rank that cannot even enter it (librccl.so not resolvable) would leave the others
            # inside it.  So availability is agreed on FIRST — every rank probes locally (dlopen + ncclGetUniqueId, no communication)
            # and the flags are MIN-reduced — and only then does anybody call comm_init.
            ok, why, uid = 1, None, None
            try:
                uid = lib.comm_unique_id()
            except Exception as exc:
                ok, why = 0, exc
            if world > 1:
                flag = torch.tensor([ok], dtype=torch.int32, device=torch_device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            if ok:
                idt = torch.zeros(128, dtype=torch.uint8, device=torch_device)
                if rank == 0:
                    idt.copy_(torch.frombuffer(bytearray(uid), dtype=torch.uint8))
                if world > 1:
                    dist.broadcast(idt, src=0)
                try:
                    self.ctx.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, world)
                except Exception as exc:      # ncclCommInitRank itself refused (every rank is inside or past it: no one is left waiting)
                    ok, why = 0, exc
                if world > 1:                 # every rank takes the same path
                    flag = torch.tensor([ok], dtype=torch.int32, device=torch_device)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    ok = int(flag.item())
            if not ok:
                # the portable exchange instead: the same all-gather issued by torch.distributed from the library's hook (the
                # context has to sit on an explicit torch stream for that: built again)
                import sys
                if rank == 0 or why is not None:
                    print("[sharded] rank %d: native RCCL exchange unavailable (%s): falling back to the torch.distributed hook" % (rank, why), file=sys.stderr)
                del self.ctx
                self.__init__(lib, window, rank, world, dist=dist, torch_device=torch_device, ctx_kwargs=ctx_kwargs, force_hook=force_hook, exchange="hook")
                return
        elif self.exchange == "hook":     # force_hook: exercise the exchange path on a single rank (tests)
            self.ctx.set_exchange_hook(self._exchange)
        elif self.exchange == "hook_host":
            self._hsend = [torch.zeros(v.shape, dtype=v.dtype).pin_memory() for v in self._send]
            self._hrecv = [torch.zeros(v.shape, dtype=v.dtype).pin_memory() for v in self._recv]
            self._hmax = torch.zeros(1, dtype=torch.float64).pin_memory()
            self.ctx.set_exchange_hook(self._exchange_host)

    def _on_stream(self):
        import contextlib
        return self.torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def _all_gather(self, recv, send):
        """recv[r * n : (r + 1) * n] = rank r's send, on every rank."""
        if self.world == 1 or self.dist is None:
            recv.copy_(send)
        else:
            self.dist.all_gather_into_tensor(recv, send)

    def _exchange(self, which):
        try:
            with self._on_stream():
                if which == 2:      # CPU libraries: max |h_ll| over the shards, in place (the HIP library carries it in the slab)
                    if self.world > 1:
                        self.dist.all_reduce(self._max_view, op=self.dist.ReduceOp.MAX)
                else:
                    self._all_gather(self._recv[which], self._send[which])
            return 0
        except Exception as exc:      # the C side turns a non-zero return into VIO_ERR_HIP
            print("exchange hook failed:", exc)
            return 1

    def _exchange_host(self, which):
        try:
            with self._on_stream():
                if which == 2:
                    self._hmax.copy_(self._max_view)
                    if self.world > 1:
                        self.dist.all_reduce(self._hmax, op=self.dist.ReduceOp.MAX)
                    self._max_view.copy_(self._hmax)
                else:
                    hs, hr = self._hsend[which], self._hrecv[which]
                    hs.copy_(self._send[which])     # D2H on the stream the library enqueues on: waits for the kernels before it
                    self._all_gather(hr, hs)
                    self._recv[which].copy_(hr)
            return 0
        except Exception as exc:
            print("exchange hook failed:", exc)
            return 1

    def solve(self, iterations=10):
        return self.ctx.solve(iterations)

    def gn_iteration(self, lam):
        self.ctx.gn_iteration(lam)

    def marginalize(self, kind):
        """Problem::Marginalize over the shards: every rank forms the partial Schur system of
        its own frame-0-hosted landmarks, the same exchange as a linearisation's sums them in rank order, and every rank runs the
        identical eigen-decomposition tail on the identical 171x171 system — the new prior needs no broadcast."""
        return self.ctx.marginalize(kind)

    def gather_landmarks(self):
        """All ranks' inverse depths, in the original landmark order."""
        local = self.ctx.get_landmarks()
        if self.world == 1:
            return local
        parts = [None] * self.world
        self.dist.all_gather_object(parts, local)
        return np.concatenate(parts)
