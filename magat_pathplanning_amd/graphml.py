"""GraphFilterBatchAttentional with the reference's constructor / addGSO / forward /
returnAttentionGSO surface (utils/graphUtils/graphML.py:4506-4685) on top of the gfx950 kernels.

Inference (no autograd) runs the HIP path through the C ABI; there is no CPU fallback: a CPU
tensor under no_grad raises.  When autograd is required (training)
the layer evaluates the same algebra with differentiable torch ops on whatever device the
tensors live on -- that composite exists for backward only and is never used for inference.  On GPU tensors the
training forward and backward of the graph part run on the HIP kernels too (_GatTrainFunction); the composite then only
serves CPU tensors (the gloo / host tests) and requests for the dense attention tensor under autograd.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import _native as nat

ZERO_TOLERANCE = 1e-9
_MODES = {"KeyQuery": nat.MODE_KEYQUERY, "GAT_modified": nat.MODE_GAT_MODIFIED, "GAT_origin": nat.MODE_GAT_ORIGIN}


class _Scratch:
    """Per-module device scratch (packed weights, workspace); never pickled."""

    def __init__(self):
        self.packed = None
        self.packed_key = None
        self.workspace = None
        self.x_scale = 0.0           # power-of-two activation scale of the layer input (0 = none); float word [4] of the status block
        self.x_scale_written = None
        self.csr = None              # CsrStructure built in the forward when addGSO did not hand one over


def _workspace(sc, need, dev):
    """Caller-owned workspace of a layer.  Its first 256 bytes are the range-guard status block (magat_gat_read_status):
    zeroed once, here, when the buffer is (re)allocated."""
    if sc.workspace is None or sc.workspace.numel() < need or sc.workspace.device != dev:
        sc.workspace = torch.empty(need, dtype=torch.uint8, device=dev)
        sc.workspace[:256].zero_()
        sc.x_scale_written = None
    if sc.x_scale_written != sc.x_scale:        # activation scale of the layer input (magat_hip.h "Activation scales")
        sc.workspace[16:20].view(torch.float32).fill_(float(sc.x_scale))
        sc.x_scale_written = sc.x_scale
    return sc.workspace


def _param_key(*tensors):
    return tuple((t.data_ptr(), t._version, str(t.device)) for t in tensors)


def _packed_weights(layer, dev, stream, G, F, K, P, mode):
    lib = nat.lib()
    sc = layer._scratch
    tensors = layer._pack_tensors()          # (weight, weight_bias | None, mixer, taps)
    key = _param_key(*[t for t in tensors if t is not None]) + (str(dev),)
    if sc.packed is None or sc.packed_key != key:
        nfl = lib.magat_gat_packed_floats(G, F, K, P, mode)
        if nfl == 0:
            raise nat.MagatNativeError("bad GAT shape G=%d F=%d K=%d P=%d" % (G, F, K, P))
        sc.packed = torch.empty(nfl, dtype=torch.float32, device=dev)
        w = [None if t is None else t.detach().to(dev, torch.float32).contiguous() for t in tensors]
        nat.check(lib.magat_gat_pack_weights(nat.ptr(w[0]), nat.ptr(w[1]), nat.ptr(w[2]), nat.ptr(w[3]),
                                             nat.ptr(sc.packed), G, F, K, P, mode, stream), "magat_gat_pack_weights")
        sc.packed_key = key
    return sc.packed


def dense_gso_to_csr(S3, self_loops=False):
    """(B,N,N) device GSO -> (rowptr int32 [B*(N+1)] absolute offsets, colidx int32 [nnz], nnz) with the
    reference's edge rule |S| > 1e-9 (graphML.py:1274-1276), or |float(S) + I| > 1e-9 for GAT_origin
    (graphML.py:1018).  Two HIP kernels + one torch cumsum."""
    lib = nat.lib()
    B, N, _ = S3.shape
    dev = S3.device
    f64 = 1 if S3.dtype == torch.float64 else 0
    with torch.cuda.device(dev):
        stream = nat.current_stream(dev)
        deg = torch.empty(B * N, dtype=torch.int32, device=dev)
        sl = int(self_loops)          # edge rule: 0 |S|>1e-9, 1 GAT_origin (S + I), 2 float(S) != 0 (GraphFilterBatch)
        nat.check(lib.magat_gso_row_degrees(nat.ptr(S3), f64, sl, nat.ptr(deg), B, N, stream), "magat_gso_row_degrees")
        ends64 = torch.cumsum(deg, 0, dtype=torch.int64)
        nnz = int(ends64[-1].item())
        if nnz >= 2 ** 31:
            raise nat.MagatNativeError("GSO with %d edges: the CSR index arrays are int32" % nnz)
        ends = ends64.to(torch.int32)
        starts = (ends - deg).contiguous()
        colidx = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
        nat.check(lib.magat_gso_fill_csr(nat.ptr(S3), f64, sl, nat.ptr(starts), nat.ptr(colidx), B, N, stream),
                  "magat_gso_fill_csr")
        rowptr = torch.empty(B, N + 1, dtype=torch.int32, device=dev)
        rowptr[:, :N] = starts.view(B, N)
        rowptr[:, N] = ends.view(B, N)[:, N - 1]
    return rowptr.reshape(-1).contiguous(), colidx, nnz


class CsrStructure:
    """CSR + CSC edge structure of one GSO tensor, made on the device by magat_gso_csr_build (N <= 1024): one streaming
    pass over S (with addGSO's scrub fused in) + one small kernel, on a side stream under the per-agent CNN.

    Capacity: the index arrays hold `cap` entries - a guess (32 edges per node at first, the dense bound B*N*N at most),
    not the dense bound: the kernel drops entries beyond it and leaves the true edge count on the device, which travels to
    pinned host memory behind the build.  `ready()` - called in front of the graph layer, when the encoder's launches are
    already queued, so the wait costs no GPU time - waits for that copy, and re-builds with a larger capacity in the rare
    case the guess was too small.  The forward kernels then get the EXACT count (their attention / scratch buffers are
    sized by it)."""

    _CAP_LIMIT = 1 << 31          # entries (int32 offsets)

    def __init__(self):
        self.key = None
        self.rowptr = self.colidx = self.cscptr = self.csc = self.nnz_dev = self.nnz_host = self.ws = None
        self.cap = 0
        self.side = self.event = None
        self.args = None
        self.nnz = None

    @staticmethod
    def supported(B, N):
        return nat.lib().magat_gso_csr_workspace_bytes(B, N) > 0 and B * N * N < CsrStructure._CAP_LIMIT

    def _alloc(self, B, N, cap, dev):
        lib = nat.lib()
        if self.rowptr is None or self.rowptr.numel() != B * (N + 1) or self.rowptr.device != dev:
            self.rowptr = torch.empty(B * (N + 1), dtype=torch.int32, device=dev)
            self.cscptr = torch.empty(B * (N + 1), dtype=torch.int32, device=dev)
            self.nnz_dev = torch.zeros(1, dtype=torch.int64, device=dev)
            self.nnz_host = torch.zeros(1, dtype=torch.int64).pin_memory()
            self.ws = torch.empty(lib.magat_gso_csr_workspace_bytes(B, N), dtype=torch.uint8, device=dev)
            self.cap = 0
        if self.cap < cap or self.colidx is None or self.colidx.device != dev:
            self.colidx = torch.empty(cap, dtype=torch.int32, device=dev)
            self.csc = torch.empty(2, cap, dtype=torch.int32, device=dev)
            self.cap = cap

    def build(self, S3, rule, scrub_nan=0, gso_mode=0, min_cap=0):
        """S3 (B,N,N) contiguous f32|f64 device tensor (scrubbed IN PLACE when asked to: the caller's stream is then
        ordered behind the pass, so that S is the scrubbed tensor for every later reader, as after the reference's addGSO)."""
        lib = nat.lib()
        B, N, _ = S3.shape
        dev = S3.device
        dense = B * N * N
        with torch.cuda.device(dev):
            if self.rowptr is not None and (self.rowptr.numel() != B * (N + 1) or self.rowptr.device != dev):
                self.cap = 0
            cap = max(self.cap, min(dense, max(int(min_cap), 32 * B * N, 1 << 12)))
            cur = torch.cuda.current_stream(dev)
            # On a side stream by default (MAGAT_CSR_SIDE=0: in the caller's stream): the build needs nothing but S, so it may
            # run under the per-agent CNN wherever the encoder kernels leave compute units free; the layer's CSR kernels wait
            # for its event
            side = os.environ.get("MAGAT_CSR_SIDE", "1") == "1"
            if side and self.side is None:
                self.side = torch.cuda.Stream(device=dev)
            run = self.side if side else cur
            if side:
                self.side.wait_stream(cur)
            with torch.cuda.stream(run):
                self._alloc(B, N, cap, dev)          # (allocated under the stream that writes them)
                rewrites = bool(scrub_nan or gso_mode)

                def phase(ph):
                    nat.check(lib.magat_gso_csr_build_phase(
                        nat.ptr(S3), 1 if S3.dtype == torch.float64 else 0, int(scrub_nan), int(gso_mode), int(rule),
                        nat.ptr(self.rowptr), nat.ptr(self.colidx), nat.ptr(self.cscptr), nat.ptr(self.csc[0]),
                        nat.ptr(self.csc[1]), self.cap, nat.ptr(self.nnz_dev), nat.ptr(self.ws), self.ws.numel(), B, N,
                        ph, nat.current_stream(dev)), "magat_gso_csr_build_phase")

                scrubbed = None
                if side and rewrites:
                    phase(1)                         # the streaming pass over S: the only part that writes S
                    scrubbed = run.record_event()
                    phase(2)
                else:
                    phase(0)
                self.nnz_host.copy_(self.nnz_dev, non_blocking=True)
                self.event = run.record_event()
            if side:
                S3.record_stream(self.side)
                for t in (self.rowptr, self.colidx, self.cscptr, self.csc, self.nnz_dev, self.ws):
                    t.record_stream(cur)             # (read by the layer's kernels on the caller's stream)
                if scrubbed is not None:
                    # S is being rewritten: nothing of the caller's may read it earlier.  Only the streaming pass writes S, so the
                    # caller's stream (next: the per-agent CNN) waits for THAT, and the structure kernel runs beside the CNN
                    cur.wait_event(scrubbed)
        # (S._version: a torch-side in-place change of S behind addGSO makes the structure stale; the kernels' own scrub does
        #  not count - it writes through the raw pointer)
        self.key = (S3.data_ptr(), S3._version, B, N, S3.dtype, int(rule), str(dev))
        self.args = (S3, rule, scrub_nan, gso_mode)      # kept until ready(): a re-build needs S once more
        self.nnz = None
        return self

    def ready(self, dev):
        """Orders the current stream behind the build and returns the exact edge count (host wait for the count's copy;
        re-build when the capacity guess was exceeded)."""
        for _ in range(3):
            self.event.synchronize()
            self.nnz = int(self.nnz_host[0])
            if self.nnz <= self.cap:
                break
            S3, rule, scrub, gmode = self.args
            self.build(S3, rule, scrub, gmode, min_cap=self.nnz + self.nnz // 4)
        else:
            raise nat.MagatNativeError("CSR structure build did not converge (nnz %d, cap %d)" % (self.nnz, self.cap))
        torch.cuda.current_stream(dev).wait_event(self.event)
        self.args = None        # the count fits: nothing needs S again (no strong reference to ~1 GB of GSO at config 5)
        return self.nnz

    def matches(self, S3, rule):
        return self.key == (S3.data_ptr(), S3._version, S3.shape[0], S3.shape[1], S3.dtype, int(rule), str(S3.device))

    def exact_nnz(self):
        return self.nnz if self.nnz is not None else self.ready(self.rowptr.device)


def gat_forward_rows_csr(X, rowptr, colidx, nnz, layer, out=None, want_attention=False, csc=None):
    """CSR / large-graph form (any N).  X (B,N,G) device rows; rowptr int32 [B*(N+1)] absolute offsets; colidx int32.
    X float32 -> fp32 kernels; X bfloat16 -> the bf16-STORAGE kernels (BASELINE config 5: X, maps, hop states and the
    result are bf16 in HBM, fp32 arithmetic; `out`, if given, is bfloat16 too - or, with the CSC view, float32: the last
    kernel then stores the bf16-rounded result widened, the values a cast of the bf16 result would give).  nnz: edge count, or any upper bound
    the index arrays were allocated with (the kernels use it as a stride / for sizing only).  csc: optional
    (cscptr, cscsrc, cscpos) made by magat_gso_csr_build - skips the per-call transpose.
    Returns (out (B*N, ld), att (P, nnz) CSR-ordered fp32 attention or None)."""
    if not X.is_cuda:
        raise nat.MagatNativeError("the HIP GAT path needs device tensors; got %s (no CPU fallback)" % X.device)
    lib = nat.lib()
    B, N, G = X.shape
    F, K, P = layer.F, layer.K, layer.P
    mode = _MODES[layer.attentionMode]
    concat = 1 if layer.concatenate else 0
    width = P * F if concat else F
    bf16 = X.dtype == torch.bfloat16
    X = X.contiguous() if bf16 else X.contiguous().float()
    sdt = torch.bfloat16 if bf16 else torch.float32
    if csc is not None:
        ws_fn = lambda *a: lib.magat_gat_csc_workspace_bytes(*a, 1 if bf16 else 0)
    else:
        ws_fn = lib.magat_gat_csr_bf16_workspace_bytes if bf16 else lib.magat_gat_csr_workspace_bytes
    dev = X.device
    sc = layer._scratch
    with torch.cuda.device(dev):
        stream = nat.current_stream(dev)
        packed = _packed_weights(layer, dev, stream, G, F, K, P, mode)
        need = ws_fn(B, N, nnz, G, F, K, P, mode, concat)
        _workspace(sc, need, dev)
        if out is None:
            out = torch.empty(B * N, width, dtype=sdt, device=dev)
        elif out.dtype != sdt and not (bf16 and out.dtype == torch.float32 and csc is not None):
            raise TypeError("out must be %s for %s rows" % (sdt, X.dtype))
        f32out = bf16 and out.dtype == torch.float32      # the last kernel widens the bf16-rounded result itself
        att = torch.empty(P, max(nnz, 1), dtype=torch.float32, device=dev) if want_attention else None
        bias = None if layer.bias is None else layer.bias.detach().to(dev, torch.float32).reshape(-1).contiguous()
        tail = (nnz, nat.ptr(packed), nat.ptr(bias), nat.ptr(out), out.stride(0), nat.ptr(att), nat.ptr(sc.workspace),
                sc.workspace.numel(), B, N, G, F, K, P, mode, concat, stream)
        if csc is not None:
            fn = lib.magat_gat_forward_csc_bf16 if bf16 else lib.magat_gat_forward_csc_f32
            if f32out:
                fn = lib.magat_gat_forward_csc_bf16_f32out
            nat.check(fn(nat.ptr(X), nat.ptr(rowptr), nat.ptr(colidx), nat.ptr(csc[0]), nat.ptr(csc[1]), nat.ptr(csc[2]),
                         *tail), "magat_gat_forward_csc_%s" % ("bf16" if bf16 else "f32"))
        else:
            fn = lib.magat_gat_forward_csr_bf16 if bf16 else lib.magat_gat_forward_csr_f32
            nat.check(fn(nat.ptr(X), nat.ptr(rowptr), nat.ptr(colidx), *tail),
                      "magat_gat_forward_csr_%s" % ("bf16" if bf16 else "f32"))
    return out, att


def _csr_attention_to_dense(att, rowptr, colidx, nnz, B, N, P):
    """(P,nnz) CSR-ordered attention -> (B,P,1,N,N) dense, only for returnAttentionGSO() callers."""
    dev = att.device
    rp = rowptr.view(B, N + 1).long()
    deg = (rp[:, 1:] - rp[:, :-1]).reshape(-1)
    rows = torch.repeat_interleave(torch.arange(B * N, device=dev), deg)
    dense = torch.zeros(P, B * N, N, dtype=torch.float32, device=dev)
    dense[:, rows, colidx[:nnz].long()] = att[:, :nnz]
    return dense.view(P, B, N, N).permute(1, 0, 2, 3).unsqueeze(2).contiguous()


def dense_route(N, layer, want_attention=False):
    """True when the layer on graphs of N agents runs on the LDS-resident kernels: the two-launch form's tiles fit, or (beyond
    them: 128 features on 106 .. 128 agents) a one-launch kernel exists - that one returns no attention tensor."""
    lib = nat.lib()
    if lib.magat_gat_dense_supported(N, layer.G, layer.F):
        return True
    return (not want_attention and
            bool(lib.magat_gat_one_launch_supported(N, layer.G, layer.F, layer.K, _MODES[layer.attentionMode],
                                                    1 if layer.concatenate else 0)))


def gat_forward_rows(X, S, layer, out=None, want_attention=False, csr=None):
    """Kernel-facing form.  X (B,N,G) f32 contiguous device rows; S (B,N,N) or (B,1,N,N) f32|f64;
    out: optional (B*N, ld) float32 view whose first P*F|F columns receive the result.
    csr: optional CsrStructure made from this S at addGSO time (large-graph / bf16-storage path).
    Returns (out (B*N, ld) with the result in columns [0, width), aij (B,P,1,N,N) device tensor or None)."""
    if not X.is_cuda:
        raise nat.MagatNativeError("the HIP GAT path needs device tensors; got %s (no CPU fallback)" % X.device)
    lib = nat.lib()
    B, N, G = X.shape
    F, K, P = layer.F, layer.K, layer.P
    mode = _MODES[layer.attentionMode]
    concat = 1 if layer.concatenate else 0
    width = P * F if concat else F
    X = X.contiguous()
    bf16 = X.dtype == torch.bfloat16           # bf16 storage: always the CSR kernels (the LDS kernel is fp32-only)
    if X.dtype != torch.float32 and not bf16:
        X = X.float()
    S3 = S.reshape(B, N, N)
    if S3.dtype not in (torch.float32, torch.float64):
        S3 = S3.float()
    if not S3.is_contiguous():
        S3 = S3.contiguous()
    if S3.device != X.device:
        S3 = S3.to(X.device)
    dev = X.device
    sc = layer._scratch
    if bf16 or not dense_route(N, layer, want_attention) or (out is not None and out.data_ptr() % 16):
        # graph too large for the LDS-resident kernel (or bf16 storage): same layer through the CSR kernels
        rule = 1 if layer.attentionMode == "GAT_origin" else 0
        if CsrStructure.supported(B, N):
            # structure made on the device, no host synchronisation: at addGSO time (csr), or here
            if csr is None or not csr.matches(S3, rule):
                if layer._scratch.csr is None:
                    layer._scratch.csr = CsrStructure()
                csr = layer._scratch.csr.build(S3, rule)
            nnz = csr.ready(X.device)
            out, att = gat_forward_rows_csr(X, csr.rowptr, csr.colidx, nnz, layer, out=out,
                                            want_attention=want_attention, csc=(csr.cscptr, csr.csc[0], csr.csc[1]))
            aij = _csr_attention_to_dense(att, csr.rowptr, csr.colidx, nnz, B, N, P) if want_attention else None
            return out, aij
        rowptr, colidx, nnz = dense_gso_to_csr(S3, self_loops=layer.attentionMode == "GAT_origin")
        out, att = gat_forward_rows_csr(X, rowptr, colidx, nnz, layer, out=out, want_attention=want_attention)
        aij = _csr_attention_to_dense(att, rowptr, colidx, nnz, B, N, P) if want_attention else None
        return out, aij
    with torch.cuda.device(dev):
        stream = nat.current_stream(dev)
        _packed_weights(layer, dev, stream, G, F, K, P, mode)
        need = lib.magat_gat_workspace_bytes(B, N, G, F, K, P, mode, concat)
        _workspace(sc, need, dev)
        if out is None:
            out = torch.empty(B * N, width, dtype=torch.float32, device=dev)
        ldy = out.stride(0)
        aij = torch.empty(B, P, 1, N, N, dtype=torch.float32, device=dev) if want_attention else None
        bias = None if layer.bias is None else layer.bias.detach().to(dev, torch.float32).reshape(-1).contiguous()
        nat.check(lib.magat_gat_forward_planned_f32(
            nat.ptr(X), nat.ptr(S3), 1 if S3.dtype == torch.float64 else 0, nat.ptr(sc.packed), nat.ptr(bias),
            nat.ptr(out), ldy, nat.ptr(aij), nat.ptr(sc.workspace), sc.workspace.numel(),
            B, N, G, F, K, P, mode, concat, None, stream), "magat_gat_forward_planned_f32")
    return out, aij


def pack_torch(weight, weight_bias, mixer, taps, mode_name):
    """Differentiable torch twin of pack_kernel (csrc/gat_f32.hip): Bt [NC][G] and column bias [NC]."""
    if mode_name == "GAT_origin":          # taps = filterWeight (1,K); h[p,f,k,g] = h_k * W[p,0,g,f]
        P, _, F, G = weight.shape
        K = taps.shape[1]
        U = torch.einsum("k,pgf->pkfg", taps[0], weight[:, 0]).reshape(P * K * F, G)
    else:
        P, F, _, K, G = taps.shape
        U = taps[:, :, 0].permute(0, 2, 1, 3).reshape(P * K * F, G)
    if mode_name == "KeyQuery":
        Bt = torch.cat((weight[:, 0].reshape(P * G, G), U), dim=0)
        return Bt, torch.zeros(Bt.shape[0], dtype=Bt.dtype, device=Bt.device)
    W = weight[:, 0]                                            # (P,F,G)
    a1, a2 = mixer[:, 0, :F], mixer[:, 0, F:]
    v1, v2 = torch.einsum("pf,pfg->pg", a1, W), torch.einsum("pf,pfg->pg", a2, W)
    nc = (P * K * F + 2 * P + 31) // 32 * 32
    pad = torch.zeros(nc - P * K * F - 2 * P, G, dtype=U.dtype, device=U.device)
    Bt = torch.cat((U, v1, v2, pad), dim=0)
    if weight_bias is None:
        return Bt, torch.zeros(nc, dtype=U.dtype, device=U.device)
    wb = weight_bias[:, 0]
    cb = torch.cat((torch.zeros(P * K * F, dtype=U.dtype, device=U.device), (a1 * wb).sum(1), (a2 * wb).sum(1),
                    torch.zeros(nc - P * K * F - 2 * P, dtype=U.dtype, device=U.device)))
    return Bt, cb


def _gemm_nt(a, bt, dev):
    """a [M][K] @ bt[N][K]^T -> [M][N] float32 on magat_linear_f32 (conv_gemm_f32.hip); K a multiple of 4 after padding."""
    M, K = a.shape
    Nn = bt.shape[0]
    if K % 4:                      # (row strides must be multiples of 4 floats: pad the contraction with zeros)
        pad = 4 - K % 4
        a = torch.nn.functional.pad(a, (0, pad))
        bt = torch.nn.functional.pad(bt, (0, pad))
        K += pad
    a, bt = a.contiguous(), bt.contiguous()
    ldy = (Nn + 3) // 4 * 4
    y = torch.empty(M, ldy, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        nat.check(nat.lib().magat_linear_f32(nat.ptr(a), K, nat.ptr(bt), None, nat.ptr(y), ldy, M, Nn, K, 0,
                                             nat.current_stream(dev)), "magat_linear_f32")
    return y[:, :Nn]


class _GatTrainFunction(torch.autograd.Function):
    """HIP forward + backward of the graph-attention layer for training (pre-activation, per-head output)."""

    @staticmethod
    def forward(ctx, X, weight, weight_bias, mixer, taps, bias, rowptr, colidx, nnz, layer):
        lib = nat.lib()
        B, N, G = X.shape
        F, K, P = layer.F, layer.K, layer.P
        mode = _MODES[layer.attentionMode]
        dev = X.device
        M = B * N
        Xc = X.detach().contiguous().float()
        with torch.cuda.device(dev):
            stream = nat.current_stream(dev)
            packed = _packed_weights(layer, dev, stream, G, F, K, P, mode)
            nc = (lib.magat_gat_packed_floats(G, F, K, P, mode) - 0)  # total floats; NC recovered below
            NC = P * G + P * K * F if mode == nat.MODE_KEYQUERY else (P * K * F + 2 * P + 31) // 32 * 32
            Ypre = torch.empty(M, P * F, dtype=torch.float32, device=dev)
            att = torch.empty(P, max(nnz, 1), dtype=torch.float32, device=dev)
            Z = torch.empty(M, NC, dtype=torch.float32, device=dev)
            T = torch.empty(max(K - 2, 0), M, P * F, dtype=torch.float32, device=dev) if K > 2 else None
            cscptr = torch.empty(B * (N + 1), dtype=torch.int32, device=dev)
            csc = torch.empty(3, max(nnz, 1), dtype=torch.int32, device=dev)
            b1 = None if bias is None else bias.detach().reshape(-1).contiguous().float()
            nat.check(lib.magat_gat_train_forward_f32(
                nat.ptr(Xc), nat.ptr(rowptr), nat.ptr(colidx), nnz, nat.ptr(packed), nat.ptr(b1), nat.ptr(Ypre),
                nat.ptr(att), nat.ptr(Z), nat.ptr(T), nat.ptr(cscptr), nat.ptr(csc[0]), nat.ptr(csc[1]),
                nat.ptr(csc[2]), B, N, G, F, K, P, mode, stream), "magat_gat_train_forward_f32")
        ctx.layer, ctx.nnz, ctx.dims = layer, nnz, (B, N, G, F, K, P, mode, NC)
        ctx.has_bias = bias is not None
        ctx.no_wb = weight_bias is None
        ctx.save_for_backward(Xc, Z, att, T if T is not None else torch.empty(0, device=dev), rowptr, colidx, cscptr,
                              csc, packed, weight, weight_bias if weight_bias is not None else torch.empty(0, device=dev),
                              mixer, taps)
        ctx.mark_non_differentiable(att)
        return Ypre, att           # att (P, nnz): the attention of every stored edge, row-major CSR order

    @staticmethod
    def backward(ctx, dYpre, _datt=None):
        lib = nat.lib()
        Xc, Z, att, T, rowptr, colidx, cscptr, csc, packed, weight, weight_bias, mixer, taps = ctx.saved_tensors
        B, N, G, F, K, P, mode, NC = ctx.dims
        dev, M = Xc.device, B * N
        dY = dYpre.contiguous().float()
        with torch.cuda.device(dev):
            stream = nat.current_stream(dev)
            dZ = torch.empty(M, NC, dtype=torch.float32, device=dev)
            dXd = torch.empty(M, G, dtype=torch.float32, device=dev)
            datt = torch.empty(P, max(ctx.nnz, 1), dtype=torch.float32, device=dev)
            nat.check(lib.magat_gat_train_backward_f32(
                nat.ptr(dY), nat.ptr(Xc), nat.ptr(Z), nat.ptr(att), nat.ptr(T if T.numel() else None), nat.ptr(rowptr),
                nat.ptr(colidx), nat.ptr(cscptr), nat.ptr(csc[0]), nat.ptr(csc[1]), ctx.nnz, nat.ptr(dZ), nat.ptr(dXd),
                nat.ptr(datt), B, N, G, F, K, P, mode, stream), "magat_gat_train_backward_f32")
        Bt = packed[:NC * G].view(NC, G)
        X2 = Xc.view(M, G)
        # the two dense products of the backward on the library's own float32 MFMA GEMM (deterministic tile order, no split-K
        # atomics): dX = dXd + dZ Bt, dBt = dZ^T X
        dX = (dXd + _gemm_nt(dZ, Bt.t().contiguous(), dev)).view(B, N, G)
        dBt, dcb = _gemm_nt(dZ.t().contiguous(), X2.t().contiguous(), dev), dZ.sum(dim=0)
        grads = [None, None, None, None]
        params = [weight, None if ctx.no_wb else weight_bias, mixer, taps]
        need = [i for i, t in enumerate(params) if t is not None and t.requires_grad]
        if need:
            with torch.enable_grad():
                leaves = [None if t is None else t.detach().requires_grad_(True) for t in params]
                Bt_t, cb_t = pack_torch(*leaves, ctx.layer.attentionMode)
                outs, gouts = [Bt_t], [dBt]
                if cb_t.requires_grad:          # KeyQuery has a constant (zero) column bias
                    outs.append(cb_t)
                    gouts.append(dcb)
                got = torch.autograd.grad(outs, [leaves[i] for i in need], gouts, allow_unused=True)
            for i, g in zip(need, got):
                grads[i] = g
        dbias = dY.view(M, P, F).sum(dim=(0, 1)).view(F, 1) if ctx.has_bias else None
        return dX, grads[0], grads[1], grads[2], grads[3], dbias, None, None, None, None


def _composite(layer, x, S):
    """Differentiable torch-op evaluation (training only).  x (B,G,N); S (B,1,N,N)."""
    B, G, N = x.shape
    P, F, K = layer.P, layer.F, layer.K
    X = x.transpose(1, 2)                                            # B,N,G
    if layer.attentionMode == "GAT_origin":
        S = S.detach().float() + torch.eye(N, dtype=torch.float32, device=S.device).view(1, 1, N, N)
    M = (S.detach().abs().sum(dim=1) > ZERO_TOLERANCE).to(x.dtype).unsqueeze(1)   # B,1,N,N
    if layer.attentionMode == "KeyQuery":
        Q = torch.einsum("bng,pog->bpno", X, layer.weight[:, 0])     # q_j = W x_j
        e = torch.einsum("big,bpjg->bpij", X, Q)
    else:
        Wx = torch.einsum("bng,pfg->bpnf", X, layer.weight[:, 0])
        if layer.attentionMode != "GAT_origin":
            Wx = Wx + layer.weight_bias[:, 0].view(1, P, 1, F)
        c1 = torch.einsum("bpnf,pf->bpn", Wx, layer.mixer[:, 0, :F])
        c2 = torch.einsum("bpnf,pf->bpn", Wx, layer.mixer[:, 0, F:])
        e = nn.functional.leaky_relu(c1.unsqueeze(2) + c2.unsqueeze(3), 0.2)
    A = torch.softmax(e * M - (1 - M) * 1e12, dim=3) * M             # B,P,N,N
    At = A.transpose(2, 3)
    Z = X.unsqueeze(1).expand(B, P, N, G)
    if layer.attentionMode == "GAT_origin":      # h[p,f,k,g] = h_k * W[p,0,g,f]
        taps = torch.einsum("k,pgf->pfkg", layer.filterWeight[0], layer.weight[:, 0])
    else:
        taps = layer.filterWeight[:, :, 0]
    y = torch.einsum("bpng,pfg->bpnf", Z, taps[:, :, 0])
    for k in range(1, K):
        Z = torch.matmul(At, Z)
        y = y + torch.einsum("bpng,pfg->bpnf", Z, taps[:, :, k])
    if layer.bias is not None:
        y = y + layer.bias.view(1, 1, 1, F)
    if layer.concatenate:
        out = torch.relu(y).permute(0, 2, 1, 3).reshape(B, N, P * F)
    else:
        out = torch.relu(y.mean(dim=1))
    return out.transpose(1, 2), A.unsqueeze(2)


def _is_relu(fn):
    return fn is nn.functional.relu or fn is torch.relu or isinstance(fn, nn.ReLU)


class _EdgeView:
    """Edge feature e of a layer as an E = 1 layer: the parameter slices (views: gradients reach the parameters) and its own
    packed-weights cache."""

    def __init__(self, layer, e):
        self.layer, self.e = layer, e
        self.F, self.K, self.P, self.G, self.attentionMode = layer.F, layer.K, layer.P, layer.G, layer.attentionMode
        self._scratch = _Scratch()

    def _pack_tensors(self):
        w, wb, mx, tp = self.layer._pack_tensors()
        e = self.e
        tp_e = tp[e:e + 1] if self.attentionMode == "GAT_origin" else tp[:, :, e:e + 1]       # filterWeight (E,K) | (P,F,E,K,G)
        return w[:, e:e + 1], None if wb is None else wb[:, e:e + 1], mx[:, e:e + 1], tp_e


class GraphFilterBatchAttentional(nn.Module):
    """Drop-in for the reference class of the same name (graphML.py:4506-4685)."""

    _edge_views = None

    def __init__(self, G, F, K, P, E=1, bias=True, nonlinearity=nn.functional.relu, concatenate=True,
                 attentionMode="GAT_modified"):
        super().__init__()
        # E > 1 and nonlinearities other than ReLU take the general path of forward() (round 4): the models use E = 1 and ReLU
        if attentionMode not in _MODES:
            raise NotImplementedError("attentionMode %r: KeyQuery and GAT_modified are built" % (attentionMode,))
        self.G, self.F, self.K, self.P, self.E = G, F, K, P, E
        self.S = None
        self.aij = None
        self.nonlinearity = nonlinearity
        self.concatenate = concatenate
        self.attentionMode = attentionMode
        self.return_attention = False      # materialise aij (B,P,E,N,N) like graphML.py:4650 only on request
        self.mixer = nn.Parameter(torch.empty(P, E, 2 * F))
        self.weight_bias = nn.Parameter(torch.empty(P, E, F))
        self.filterWeight = nn.Parameter(torch.empty(P, F, E, K, G))
        if bias:
            self.bias = nn.Parameter(torch.empty(F, 1))
        else:
            self.register_parameter("bias", None)
        if attentionMode == "KeyQuery":
            self.weight = nn.Parameter(torch.empty(P, E, G, G))
        else:
            self.weight = nn.Parameter(torch.empty(P, E, F, G))
        self._scratch = _Scratch()
        self.reset_parameters()

    # torch.float32 (default) or torch.bfloat16: HBM storage type of the node features inside the layer at inference
    # (BASELINE config 5).  bf16 always takes the CSR kernels; arithmetic stays fp32.  Not part of the state_dict.
    storage_dtype = torch.float32

    def _pack_tensors(self):
        return self.weight, self.weight_bias, self.mixer, self.filterWeight

    def reset_parameters(self):
        # graphML.py:4604-4612
        stdv = 1.0 / math.sqrt(self.G * self.P)
        with torch.no_grad():
            self.weight.uniform_(-stdv, stdv)
            self.weight_bias.zero_()
            self.mixer.uniform_(-stdv, stdv)
            self.filterWeight.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def __getstate__(self):
        st = self.__dict__.copy()
        st["_scratch"] = None
        st["aij"] = None
        st["_edge_views"] = None
        return st

    def __setstate__(self, st):
        super().__setstate__(st)
        self._scratch = _Scratch()

    def addGSO(self, S):
        assert len(S.shape) == 4
        assert S.shape[1] == self.E
        self.N = S.shape[2]
        assert S.shape[3] == self.N
        self.S = S

    def returnAttentionGSO(self):
        if self.aij is None:
            raise RuntimeError("attention was not materialised: set layer.return_attention = True before forward")
        aij = self.aij.detach().cpu().numpy() if torch.is_tensor(self.aij) else self.aij
        assert len(aij.shape) == 5 and aij.shape[2] == self.E
        return np.mean(aij, axis=1)

    def forward(self, x):
        B, Gin, Nin = x.shape
        assert self.S is not None, "addGSO must be called before forward"
        if Nin < self.N:
            x = torch.cat((x, torch.zeros(B, Gin, self.N - Nin, dtype=x.dtype, device=x.device)), dim=2)
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if self.E != 1 or not _is_relu(self.nonlinearity):
            y = self._forward_general(x)
        elif needs_grad and x.is_cuda and not self.return_attention:
            # training on the GPU: HIP forward + backward of the graph layer (CSR kernels), ReLU / head merge in torch
            N = self.N
            S3 = self.S.reshape(B, N, N).to(x.device)
            if S3.dtype not in (torch.float32, torch.float64):
                S3 = S3.float()
            rowptr, colidx, nnz = dense_gso_to_csr(S3.contiguous(), self_loops=self.attentionMode == "GAT_origin")
            w_, wb_, mx_, tp_ = self._pack_tensors()
            Ypre, _ = _GatTrainFunction.apply(x.permute(0, 2, 1).contiguous(), w_, wb_, mx_, tp_, self.bias, rowptr,
                                              colidx, nnz, self)
            Yh = Ypre.view(B, N, self.P, self.F)
            if self.concatenate:
                y = torch.relu(Yh).reshape(B, N, self.P * self.F).permute(0, 2, 1)
            else:
                y = torch.relu(Yh.mean(dim=2)).permute(0, 2, 1)
            self.aij = None
        elif needs_grad:
            nat.require_device_or_composite(x, "GraphFilterBatchAttentional under autograd")
            y, aij = _composite(self, x, self.S.to(x.device))
            self.aij = aij.detach() if self.return_attention else None
        else:
            rows = x.permute(0, 2, 1).contiguous()
            if self.storage_dtype == torch.bfloat16:
                rows = rows.to(torch.bfloat16)
            out, aij = gat_forward_rows(rows, self.S, self, want_attention=self.return_attention)
            self.aij = aij
            y = out.reshape(B, self.N, out.shape[1]).permute(0, 2, 1).float()
        if Nin < self.N:
            y = y[:, :, :Nin]
        return y

    def _forward_general(self, x):
        """E > 1 edge features and / or a nonlinearity other than ReLU (graphML.py:1262-1286, 1744-1775, 4654-4667) on the HIP
        kernels of the training path, which hand out the PRE-activation per-head rows: the edge mask is the union of the E
        GSOs (sum_e |S_e| > 1e-9 - for GAT_origin of |float(S_e) + I|), every edge feature is an E = 1 layer over that mask
        with its own slice of the parameters (magat_gat_train_forward_f32), their rows are summed, the bias is added once and
        the constructor's nonlinearity - any torch callable - is applied in the reference's own layout.  Differentiable: the
        slices are views of the parameters and each E = 1 pass carries the HIP backward."""
        if not x.is_cuda:
            raise nat.MagatNativeError("GraphFilterBatchAttentional (E > 1 or a nonlinearity other than ReLU) runs on the HIP "
                                       "kernels: move the module and its input to the GPU")
        B, _, N = x.shape
        P, F, E = self.P, self.F, self.E
        S = self.S.to(x.device)
        if S.dtype not in (torch.float32, torch.float64):
            S = S.float()
        origin = self.attentionMode == "GAT_origin"
        if E == 1:
            Su, loops = S.reshape(B, N, N), origin
        elif origin:
            Su, loops = (S.float() + torch.eye(N, dtype=torch.float32, device=S.device).view(1, 1, N, N)).abs().sum(dim=1), False
        else:
            Su, loops = S.abs().sum(dim=1), False
        rowptr, colidx, nnz = dense_gso_to_csr(Su.contiguous(), self_loops=loops)
        rows = x.permute(0, 2, 1).contiguous()
        if self._edge_views is None or len(self._edge_views) != E:
            self._edge_views = [_EdgeView(self, e) for e in range(E)]
        Ypre, atts = None, []
        for e, view in enumerate(self._edge_views):
            w_, wb_, mx_, tp_ = view._pack_tensors()
            Ye, att = _GatTrainFunction.apply(rows, w_, wb_, mx_, tp_, self.bias if e == 0 else None, rowptr, colidx, nnz, view)
            Ypre = Ye if Ypre is None else Ypre + Ye
            atts.append(att)
        if self.return_attention:
            rp = rowptr.view(B, N + 1)
            deg = (rp[:, 1:] - rp[:, :-1]).reshape(-1).long()
            r = torch.repeat_interleave(torch.arange(B * N, device=x.device), deg)
            A = torch.zeros(P, E, B * N, N, dtype=torch.float32, device=x.device)
            for e, att in enumerate(atts):
                A[:, e, r, colidx[:nnz].long()] = att[:, :nnz]
            self.aij = A.view(P, E, B, N, N).permute(2, 0, 1, 3, 4).contiguous()
        else:
            self.aij = None
        y = Ypre.view(B, N, P, F).permute(0, 2, 3, 1)                  # B x P x F x N, as graphML.py:4650 hands it over
        if self.concatenate:
            y = self.nonlinearity(y)
            return y.permute(0, 3, 1, 2).reshape(B, N, P * F).permute(0, 2, 1)
        return self.nonlinearity(torch.mean(y, dim=1))

    def extra_repr(self):
        s = "in_features=%d, out_features=%d, filter_taps=%d, attention_heads=%d, edge_features=%d, bias=%s, " % (
            self.G, self.F, self.K, self.P, self.E, self.bias is not None)
        s += "attentionMode=%s, " % self.attentionMode
        s += ("GSO stored: number_nodes=%d" % self.N) if self.S is not None else "no GSO stored"
        return s


class GraphFilterBatchAttentional_Origin(GraphFilterBatchAttentional):
    """Drop-in for the reference class of the same name (graphML.py:4175-4339), attentionMode 'GAT_origin': the
    GAT baseline of the paper.  Parameters: mixer (P,E,2F), weight (P,E,F,G), filterWeight (E,K) scalar taps, bias (F,1).
    Self-loops are added to the GSO; the filter is h_k * W (transposed, see oracle.attention notes).  Same kernels as
    GAT_modified with a different weight packing and mask rule (MAGAT_MODE_GAT_ORIGIN)."""

    def __init__(self, G, F, K, P, E=1, bias=True, nonlinearity=nn.functional.relu, concatenate=True,
                 attentionMode="GAT_origin"):
        nn.Module.__init__(self)
        if G != F:
            raise NotImplementedError("GAT_origin needs F == G (the reference reshapes W (P,G,E,F) into (P,F,E,1,G))")
        self.G, self.F, self.K, self.P, self.E = G, F, K, P, E
        self.S = None
        self.aij = None
        self.nonlinearity = nonlinearity
        self.concatenate = concatenate
        self.attentionMode = "GAT_origin"
        self.return_attention = False
        self.mixer = nn.Parameter(torch.empty(P, E, 2 * F))
        self.weight = nn.Parameter(torch.empty(P, E, F, G))
        self.filterWeight = nn.Parameter(torch.empty(E, K))
        if bias:
            self.bias = nn.Parameter(torch.empty(F, 1))
        else:
            self.register_parameter("bias", None)
        self._scratch = _Scratch()
        self.reset_parameters()

    def _pack_tensors(self):
        return self.weight, None, self.mixer, self.filterWeight

    def reset_parameters(self):
        # graphML.py:4259-4266
        stdv = 1.0 / math.sqrt(self.G * self.P)
        with torch.no_grad():
            self.weight.uniform_(-stdv, stdv)
            self.mixer.uniform_(-stdv, stdv)
            self.filterWeight.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)


class _GnnTrainFunction(torch.autograd.Function):
    """HIP forward + backward of GraphFilterBatch for training.  The layer is linear, so nothing but the input rows and the
    CSR arrays is kept:  dU_k = (A^T)^k dY  by the HIP hop kernel (magat_gnn_backward_csr_f32), then two library GEMMs."""

    @staticmethod
    def forward(ctx, x, weight, bias, layer):
        y, X, (rowptr, colidx, vals, nnz) = layer._forward_hip(x)
        ctx.layer, ctx.nnz, ctx.shape = layer, nnz, tuple(x.shape)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(X, rowptr, colidx, vals, weight)
        # (a tensor of its own: _forward_hip hands out a permuted VIEW of the kernel's rows, and an in-place op behind the layer -
        #  the model's ReLU(inplace=True) - on a view made inside a custom Function is refused by autograd)
        return y.clone()

    @staticmethod
    def backward(ctx, dy):
        X, rowptr, colidx, vals, weight = ctx.saved_tensors
        layer = ctx.layer
        B, G, N = ctx.shape
        F, K = layer.F, layer.K
        dev, M = X.device, B * N
        dY = dy.permute(0, 2, 1).contiguous().float().view(M, F)
        dZ = torch.empty(M, K * F, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            nat.check(nat.lib().magat_gnn_backward_csr_f32(nat.ptr(dY), nat.ptr(rowptr), nat.ptr(colidx), nat.ptr(vals),
                                                           ctx.nnz, nat.ptr(dZ), B, N, F, K, nat.current_stream(dev)),
                      "magat_gnn_backward_csr_f32")
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            Bt = weight.detach()[:, 0].permute(1, 0, 2).reshape(K * F, G).float()        # row k*F+f = weight[f,0,k,:]
            dx = _gemm_nt(dZ, Bt.t().contiguous(), dev).reshape(B, N, G).permute(0, 2, 1)
        if ctx.needs_input_grad[1]:
            dw = _gemm_nt(dZ.t().contiguous(), X.t().contiguous(), dev).reshape(K, F, G).permute(1, 0, 2) \
                .reshape(F, 1, K, G).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dY.sum(dim=0).view(F, 1)
        return dx, dw, db, None


class GraphFilterBatch(nn.Module):
    """Drop-in for the reference's non-attentional graph filter (graphML.py:5581-5700; BatchLSIGF :5485-5579), the GNN
    baseline of the paper:  y = bias + sum_k (x S^k) h_k  with the GSO VALUES as edge weights (`x @ S.float()`), no
    nonlinearity inside.  Parameters: weight (F,E,K,G), bias (F,1); init U(+-1/sqrt(G K)).  Inference on the HIP CSR kernels
    (magat_gnn_forward_csr_f32); with autograd on, the same forward plus the HIP backward (_GnnTrainFunction) on device
    tensors, and a torch composite for CPU tensors (host / gloo tests)."""

    def __init__(self, G, F, K, E=1, bias=True):
        super().__init__()
        if E != 1:
            raise NotImplementedError("edge_features E=1 only")
        self.G, self.F, self.K, self.E = G, F, K, E
        self.S = None
        self.weight = nn.Parameter(torch.empty(F, E, K, G))
        if bias:
            self.bias = nn.Parameter(torch.empty(F, 1))
        else:
            self.register_parameter("bias", None)
        self._scratch = _Scratch()
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.G * self.K)       # graphML.py:5654-5659
        with torch.no_grad():
            self.weight.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def addGSO(self, S):
        assert len(S.shape) == 4 and S.shape[1] == self.E and S.shape[2] == S.shape[3]     # graphML.py:5661-5668
        self.N = S.shape[2]
        self.S = S

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_scratch"] = _Scratch()
        return d

    def forward(self, x):
        B, Gin, Nin = x.shape
        assert self.S is not None, "addGSO must be called before forward"
        N = self.N
        if Nin < N:
            x = torch.cat((x, torch.zeros(B, Gin, N - Nin, dtype=x.dtype, device=x.device)), dim=2)
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs_grad and not x.is_cuda:
            nat.require_device_or_composite(x, "GraphFilterBatch under autograd")     # CPU tensors: the torch composite
            Sf = self.S.to(x.device)[:, 0].to(x.dtype)      # (the checker runs this composite in float64 too)
            z = x
            y = torch.einsum("bgn,fg->bfn", z, self.weight[:, 0, 0])
            for k in range(1, self.K):
                z = torch.matmul(z, Sf)
                y = y + torch.einsum("bgn,fg->bfn", z, self.weight[:, 0, k])
            if self.bias is not None:
                y = y + self.bias
        elif needs_grad:
            y = _GnnTrainFunction.apply(x, self.weight, self.bias, self)
        else:
            y = self._forward_hip(x)[0]
        if Nin < N:
            y = y[:, :, :Nin]
        return y

    def _csr(self, dev, B):
        """CSR arrays of float(S) (every non-zero entry, values kept) for the HIP kernels."""
        N = self.N
        S3 = self.S.reshape(B, N, N).to(dev)
        if S3.dtype not in (torch.float32, torch.float64):
            S3 = S3.float()
        S3 = S3.contiguous()
        rowptr, colidx, nnz = dense_gso_to_csr(S3, self_loops=2)          # rule 2: every non-zero float(S)
        rp = rowptr.view(B, N + 1).long()
        rows = torch.repeat_interleave(torch.arange(B * N, device=dev), (rp[:, 1:] - rp[:, :-1]).reshape(-1))
        vals = S3.reshape(B * N, N)[rows, colidx[:nnz].long()].float().contiguous() if nnz else \
            torch.zeros(1, dtype=torch.float32, device=dev)
        return rowptr, colidx, vals, nnz

    def _forward_hip(self, x):
        """x (B,G,N) device tensor -> (y (B,F,N) view, X rows (M,G), csr) on the HIP CSR kernels."""
        if not x.is_cuda:
            raise nat.MagatNativeError("the HIP path needs device tensors; got %s (no CPU fallback)" % x.device)
        lib = nat.lib()
        dev = x.device
        B, _, N = x.shape
        X = x.detach().permute(0, 2, 1).contiguous().float()
        rowptr, colidx, vals, nnz = csr = self._csr(dev, B)
        sc = self._scratch
        with torch.cuda.device(dev):
            stream = nat.current_stream(dev)
            key = _param_key(self.weight) + (str(dev),)
            if sc.packed is None or sc.packed_key != key:
                nfl = lib.magat_gat_packed_floats(self.G, self.F, self.K, 1, nat.MODE_GNN)
                sc.packed = torch.empty(nfl, dtype=torch.float32, device=dev)
                w = self.weight.detach().to(dev, torch.float32).contiguous()
                nat.check(lib.magat_gat_pack_weights(None, None, None, nat.ptr(w), nat.ptr(sc.packed), self.G, self.F,
                                                     self.K, 1, nat.MODE_GNN, stream), "magat_gat_pack_weights")
                sc.packed_key = key
            need = lib.magat_gat_csr_workspace_bytes(B, N, nnz, self.G, self.F, self.K, 1, nat.MODE_GNN, 1)
            _workspace(sc, need, dev)
            out = torch.empty(B * N, self.F, dtype=torch.float32, device=dev)
            bias = None if self.bias is None else self.bias.detach().to(dev, torch.float32).reshape(-1).contiguous()
            nat.check(lib.magat_gnn_forward_csr_f32(
                nat.ptr(X), nat.ptr(rowptr), nat.ptr(colidx), nat.ptr(vals), nnz, nat.ptr(sc.packed), nat.ptr(bias),
                nat.ptr(out), out.stride(0), nat.ptr(sc.workspace), sc.workspace.numel(), B, N, self.G, self.F,
                self.K, stream), "magat_gnn_forward_csr_f32")
        return out.view(B, N, self.F).permute(0, 2, 1), X.view(B * N, self.G), csr

    def extra_repr(self):
        return "in_features=%d, out_features=%d, filter_taps=%d, edge_features=%d, bias=%s, GSO stored: %s" % (
            self.G, self.F, self.K, self.E, self.bias is not None, self.S is not None)
