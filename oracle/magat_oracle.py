"""CPU oracle for MAGAT's batched graph-attention forward.  TEST INFRASTRUCTURE ONLY.

This is a functional restatement (torch-CPU + a loop-level numpy cross-check) of the
reference path

    DecentralPlannerGATNet.forward   graphs/models/decentralplanner_GAT_bottleneck*.py:280-341
    DecentralPlannerGATNet.addGSO    graphs/models/decentralplanner_GAT_bottleneck.py:262-278
    GraphFilterBatchAttentional      utils/graphUtils/graphML.py:4506-4685
    graphAttentionLSIGFBatch_*       utils/graphUtils/graphML.py:1724-1827
    learnAttentionGSOBatch_KeyQuery  utils/graphUtils/graphML.py:1180-1286
    learnAttentionGSOBatch           utils/graphUtils/graphML.py:713-823
    ResNet / ResNetSlim / BasicBlock graphs/models/resnet_pytorch.py:40-73,334-524

It keeps the reference's *dense* op sequence (materialised (B,P,N,N) attention, hop
stacking, un-folded BatchNorm) so that, timed on host cores, it is a cost-equivalent
stand-in for the reference's CPU forward (bench.py `cpu_baseline`, kind "port").

Pinning: oracle/make_golden.py imports the real reference (build container only) and
writes tests/golden/*.npz; tests/test_oracle_golden.py checks every function here
against those vectors.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module -- the product (magat_pathplanning_amd) never does.
"""
import math

import numpy as np
import torch
import torch.nn.functional as tnf

ZERO_TOLERANCE = 1e-9   # graphML.py:45
INFINITE_NUMBER = 1e12  # graphML.py:46

SKIP_MODES = ("BottomNeck_only", "BottomNeck_skipConcat", "BottomNeck_skipConcatGNN",
              "BottomNeck_skipAddGNN")


# --------------------------------------------------------------------------- GSO
def add_gso(S, gso_mode="dist_GSO", bottleneck_mode="BottomNeck_only"):
    """addGSO (decentralplanner_GAT_bottleneck.py:262-278).  Mutates S in place like
    the reference; returns the (B,1,N,N) tensor the layer sees.  NaN scrubbing exists
    only in the plain bottleneck file (the Skip* variants omit it)."""
    assert S.dim() == 3
    S4 = S.unsqueeze(1)
    if bottleneck_mode in ("BottomNeck_only", ""):
        S4[torch.isnan(S4)] = 0
    if gso_mode == "dist_GSO_one":
        S4[S4 > 0] = 1
    elif gso_mode == "full_GSO":
        S4 = torch.ones_like(S4)
    return S4


def edge_mask(S4, dtype=torch.float32):
    """mask = (sum_e |S| > 1e-9) cast to x.dtype   (graphML.py:1274-1276, 808-810)."""
    B, E, N, _ = S4.shape
    return (S4.abs().sum(dim=1) > ZERO_TOLERANCE).to(dtype).reshape(B, 1, 1, N, N)


# ------------------------------------------------------------------ attention GSO
def attention_keyquery(x, W, S4):
    """e_ij = x_i^T W_p x_j, masked row softmax (graphML.py:1180-1286).
    x (B,G,N); W (P,E,G,G); S4 (B,E,N,N) -> (B,P,E,N,N)."""
    B, G, N = x.shape
    P, E = W.shape[0], W.shape[1]
    xq = x.reshape(B, 1, 1, G, N)
    xk = xq.transpose(3, 4)
    Wx = torch.matmul(W.reshape(1, P, E, G, G), xq)
    e = torch.matmul(xk, Wx)
    m = edge_mask(S4, x.dtype)
    a = torch.softmax(e * m - (1 - m) * INFINITE_NUMBER, dim=4)
    return a * m


def attention_modified(x, mixer, W, Wb, S4, negative_slope=0.2):
    """e_ij = lrelu(a1.(W x_j + wb) + a2.(W x_i + wb)) (graphML.py:713-823).
    x (B,G,N); mixer (P,E,2F); W (P,E,F,G); Wb (P,E,F)."""
    B, G, N = x.shape
    P, E, F = W.shape[0], W.shape[1], W.shape[2]
    Wx = torch.matmul(W.reshape(1, P, E, F, G), x.reshape(B, 1, 1, G, N))
    Wx = Wx + Wb.reshape(1, P, E, F, 1)
    a1 = mixer[:, :, :F].reshape(1, P, E, 1, F)
    a2 = mixer[:, :, F:].reshape(1, P, E, 1, F)
    row = torch.matmul(a1, Wx)                    # B,P,E,1,N   indexed by j
    col = torch.matmul(a2, Wx).transpose(3, 4)    # B,P,E,N,1   indexed by i
    e = tnf.leaky_relu(row + col, negative_slope)
    m = edge_mask(S4, x.dtype)
    a = torch.softmax(e * m - (1 - m) * INFINITE_NUMBER, dim=4)
    return a * m


def attention_origin(x, mixer, W, S4, negative_slope=0.2):
    """GAT_origin scores (graphML.py:964-1069): self-loops added to the GSO (S.float() + I), no weight_bias."""
    B, G, N = x.shape
    P, E, F = W.shape[0], W.shape[1], W.shape[2]
    S4 = S4.float() + torch.eye(N, dtype=torch.float32).reshape(1, 1, N, N)
    Wx = torch.matmul(W.reshape(1, P, E, F, G), x.reshape(B, 1, 1, G, N))
    row = torch.matmul(mixer[:, :, :F].reshape(1, P, E, 1, F), Wx)
    col = torch.matmul(mixer[:, :, F:].reshape(1, P, E, 1, F), Wx).transpose(3, 4)
    e = tnf.leaky_relu(row + col, negative_slope)
    m = edge_mask(S4, x.dtype)
    a = torch.softmax(e * m - (1 - m) * INFINITE_NUMBER, dim=4)
    return a * m


def lsigf_attention(h, x, aij, b):
    """K-tap filter with the learned attention as shift (graphML.py:1744-1775):
    z_0 = x, z_k = z_{k-1} @ aij (column aggregation), y = sum_k z_k h_k + b.
    h (P,F,E,K,G); x (B,G,N); aij (B,P,E,N,N) -> y (B,P,F,N)."""
    P, F, E, K, G = h.shape
    B, _, N = x.shape
    cur = x.reshape(B, 1, 1, G, N)
    taps = [cur.expand(B, P, E, G, N)]
    for _ in range(1, K):
        cur = torch.matmul(cur, aij)
        taps.append(cur)
    z = torch.stack(taps, dim=3)                                   # B,P,E,K,G,N
    z = z.permute(0, 1, 5, 2, 3, 4).reshape(B, P, N, E * K * G)
    hh = h.reshape(1, P, F, E * K * G).transpose(2, 3)
    y = torch.matmul(z, hh).transpose(2, 3)                        # B,P,F,N
    if b is not None:
        y = y + b
    return y


def gat_layer_forward(x, S4, p, mode="KeyQuery", concat=True, n_graph=None, nonlinearity=torch.relu):
    """GraphFilterBatchAttentional.forward (graphML.py:4636-4671).
    x (B,G,Nin) f32; S4 (B,E,N,N); p: dict with mixer, weight_bias, filterWeight, bias,
    weight (torch tensors).  Returns (y (B,P*F|F,Nin), aij (B,P,E,N,N)).  E > 1: the edge mask is the union of the E GSOs
    (edge_mask), scores and taps are per (head, edge feature), the taps of all edge features are summed (lsigf_attention);
    nonlinearity: the layer's constructor argument, applied to (B,P,F,N) before the concat or to the head mean (B,F,N)."""
    B, G, Nin = x.shape
    N = S4.shape[2] if n_graph is None else n_graph
    if Nin < N:
        x = torch.cat((x, torch.zeros(B, G, N - Nin, dtype=x.dtype)), dim=2)
    taps = p["filterWeight"]
    if mode == "KeyQuery":
        aij = attention_keyquery(x, p["weight"], S4)
    elif "GAT_modified" in mode:
        aij = attention_modified(x, p["mixer"], p["weight"], p["weight_bias"], S4)
    elif mode == "GAT_origin":
        # GraphFilterBatchAttentional_Origin (graphML.py:4175-4339): scalar taps (E,K) times W (graphML.py:1967-1970)
        aij = attention_origin(x, p["mixer"], p["weight"], S4)
        W = p["weight"]                                              # (P,E,F,G)
        Pn, En, Fn, Gn = W.shape
        # the reference builds "P x F x E x G" with permute(0,3,1,2) -- which is (P,G,E,F) -- and then RESHAPES it to
        # (P,F,E,1,G) (graphML.py:1967-1969): with F == G the filter uses W transposed, h[p,f,k,g] = hk * W[p,0,g,f]
        Wq = W.permute(0, 3, 1, 2).reshape(Pn, Fn, En, 1, Gn)
        taps = taps.reshape(1, 1, En, -1, 1) * Wq                    # (P,F,E,K,G)
    else:
        raise ValueError("oracle covers KeyQuery, GAT_modified and GAT_origin, got %r" % mode)
    y = lsigf_attention(taps, x, aij, p.get("bias"))
    P, F = taps.shape[0], taps.shape[1]
    if concat:
        y = nonlinearity(y)
        y = y.permute(0, 3, 1, 2).reshape(B, N, P * F).permute(0, 2, 1)
    else:
        y = nonlinearity(y.mean(dim=1))
    if Nin < N:
        y = y[:, :, :Nin]
    return y, aij


def _r16(t):
    """round-to-nearest-even to bfloat16, carried as float32"""
    return t.to(torch.bfloat16).to(torch.float32)


def gat_layer_forward_bf16_storage(x, S4, p, mode="KeyQuery", concat=True):
    """The layer with bf16 STORAGE of the node features (BASELINE config 5).  NO REFERENCE COUNTERPART: the reference
    has no bf16 path, so this is gat_layer_forward's algebra in the Horner order the kernels use
    (Y = U_0 + A^T(U_1 + A^T U_2), U_k = X H_k^T) with RNE-bf16 rounding inserted exactly where the HIP bf16 path stores
    to HBM: the input rows X, the packed score/tap weights, the hoisted maps Z = [Q | U | c1 c2], every hop state and
    the result.  Arithmetic in float32, attention fp32.  The bf16 path is judged (a) as error against the pinned fp32
    oracle and (b) for agreement with this emulation to ~1 bf16 ulp.
    x (B,G,N) f32; returns (y (B,P*F|F,N) f32 holding bf16-representable values, aij (B,P,1,N,N))."""
    B, G, N = x.shape
    X = _r16(x.permute(0, 2, 1).float())                                  # (B,N,G)
    W = p["weight"].float()
    P = W.shape[0]
    bias = p.get("bias")
    if mode == "GAT_origin":
        fw = p["filterWeight"].float().reshape(-1)
        taps = torch.einsum("k,pgf->pfkg", fw, W[:, 0])                    # (P,F,K,G)
    else:
        taps = p["filterWeight"].float()[:, :, 0]                          # (P,F,K,G)
    F, K = taps.shape[1], taps.shape[2]
    U = _r16(torch.einsum("bng,pfkg->bpknf", X, _r16(taps)))               # bias-free maps, stored bf16
    mask = edge_mask(S4 if mode != "GAT_origin" else S4.float() + torch.eye(N).view(1, 1, N, N), torch.float32)[:, 0, 0]
    if mode == "KeyQuery":
        Q = _r16(torch.einsum("bng,phg->bpnh", X, _r16(W[:, 0])))          # Q[j] = W x_j
        e = torch.einsum("big,bpjg->bpij", X, Q)
    else:
        a1, a2 = p["mixer"].float()[:, 0, :F], p["mixer"].float()[:, 0, F:]
        v1, v2 = _r16(torch.einsum("pf,pfg->pg", a1, W[:, 0])), _r16(torch.einsum("pf,pfg->pg", a2, W[:, 0]))
        if mode == "GAT_origin":
            cb1 = cb2 = torch.zeros(P)
        else:
            wb = p["weight_bias"].float()[:, 0]
            cb1, cb2 = (a1 * wb).sum(1), (a2 * wb).sum(1)
        c1 = _r16(torch.einsum("bng,pg->bpn", X, v1) + cb1.view(1, P, 1))
        c2 = _r16(torch.einsum("bng,pg->bpn", X, v2) + cb2.view(1, P, 1))
        e = tnf.leaky_relu(c1.unsqueeze(2) + c2.unsqueeze(3), 0.2)          # e[i,j] = lrelu(c1[j] + c2[i])
    m4 = mask.unsqueeze(1)
    aij = torch.softmax(e * m4 - (1 - m4) * INFINITE_NUMBER, dim=3) * m4     # (B,P,N,N)
    At = aij.transpose(2, 3)
    T = U[:, :, K - 1]
    for k in range(K - 2, -1, -1):
        T = U[:, :, k] + torch.matmul(At, T)
        if k > 0:
            T = _r16(T)
    if bias is not None:
        T = T + bias.float().reshape(1, 1, 1, F)
    if concat:
        y = _r16(torch.relu(T)).permute(0, 2, 1, 3).reshape(B, N, P * F)
    else:
        y = _r16(torch.relu(_r16(T).mean(dim=1)))
    return y.permute(0, 2, 1), aij.unsqueeze(2)


def gat_layer_forward_bf16_fused(x, S4, p):
    """The bf16-storage layer in the order of the FUSED CSR kernels (csrc/gat_csr_fused.hip: KeyQuery, K = 2, concat) - which
    is the reference's own order (graphML.py:1757 hop on the node features, :1768-1770 tap contraction afterwards), not the
    Horner order of gat_layer_forward_bf16_storage.  NO REFERENCE COUNTERPART (the reference has no bf16 path): RNE-bf16
    rounding exactly where the kernels round - the input rows X, the weights, q'_i = W_p^T x_i (the registers of the score
    loop), the hop result z_j = sum_i a_ij x_i (the matrix-core operand) and the result; float32 arithmetic and attention.
    x (B,G,N) f32; returns (y (B,P*F,N) f32 holding bf16-representable values, aij (B,P,1,N,N))."""
    B, G, N = x.shape
    X = _r16(x.permute(0, 2, 1).float())                                   # (B,N,G)
    W = _r16(p["weight"].float()[:, 0])                                    # (P,G,G)
    taps = _r16(p["filterWeight"].float()[:, :, 0])                        # (P,F,K,G)
    P, F, K = taps.shape[0], taps.shape[1], taps.shape[2]
    assert K == 2
    bias = p.get("bias")
    mask = edge_mask(S4, torch.float32)[:, 0, 0]
    qp = _r16(torch.einsum("big,pgh->bpih", X, W))                         # q'_i[h] = sum_g x_i[g] W_p[g][h]
    e = torch.einsum("bpih,bjh->bpij", qp, X)
    m4 = mask.unsqueeze(1)
    aij = torch.softmax(e * m4 - (1 - m4) * INFINITE_NUMBER, dim=3) * m4     # (B,P,N,N)
    Z1 = _r16(torch.einsum("bpij,big->bpjg", aij, X))                      # z_j = sum_i a_ij x_i
    T = torch.einsum("bng,pfg->bpnf", X, taps[:, :, 0]) + torch.einsum("bpng,pfg->bpnf", Z1, taps[:, :, 1])
    if bias is not None:
        T = T + bias.float().reshape(1, 1, 1, F)
    y = _r16(torch.relu(T)).permute(0, 2, 1, 3).reshape(B, N, P * F)
    return y.permute(0, 2, 1), aij.unsqueeze(2)


def graph_filter_batch_forward(x, S4, weight, bias=None):
    """GraphFilterBatch.forward -> BatchLSIGF (graphML.py:5670-5689, 5485-5579), the non-attentional GNN baseline:
    z_0 = x, z_k = z_{k-1} @ S.float() (:5562), y = cat_k(z_k) contracted with weight (F,E,K,G) (:5573-5574), + bias.
    x (B,G,N) f32; S4 (B,1,N,N); returns (B,F,N)."""
    B, G, N = x.shape
    F, E, K, _ = weight.shape
    Sf = S4.float()
    xe = x.reshape(B, 1, G, N)
    z = x.reshape(B, 1, 1, G, N).repeat(1, E, 1, 1, 1)
    for _ in range(1, K):
        xe = torch.matmul(xe, Sf)
        z = torch.cat((z, xe.reshape(B, E, 1, G, N)), dim=2)
    y = torch.matmul(z.permute(0, 4, 1, 2, 3).reshape(B, N, E * K * G), weight.reshape(F, E * K * G).permute(1, 0)).permute(0, 2, 1)
    if bias is not None:
        y = y + bias
    return y

# ------------------------------------------------- loop-level numpy cross-check
def gat_layer_forward_loops(x, S4, p, mode="KeyQuery", concat=True):
    """Same layer, written edge-by-edge in float64 numpy straight from the formulas of
    SURVEY.md section 8(a) ("verified restatement").  Small cases only."""
    x = np.asarray(x, np.float64)
    S = np.asarray(S4, np.float64)
    B, G, N = x.shape
    W = np.asarray(p["weight"], np.float64)
    Hf = np.asarray(p["filterWeight"], np.float64)
    origin = mode == "GAT_origin"
    if origin:     # h[p,f,0,k,g] = h_k * W[p,0,g,f]  (transposed W: graphML.py:1967-1969, F == G)
        Hf = np.einsum("k,pgf->pfkg", Hf.reshape(-1), W[:, 0])[:, :, None]
    P, F, _, K, _ = Hf.shape
    bias = None if p.get("bias") is None else np.asarray(p["bias"], np.float64).reshape(F)
    X = x.transpose(0, 2, 1)                                   # B,N,G rows = nodes
    A = np.zeros((B, P, N, N))
    Y = np.zeros((B, P, N, F))
    for b in range(B):
        M = np.abs(S[b]).sum(axis=0) > ZERO_TOLERANCE
        if origin:     # self-loops: |float32(S) + I| > 1e-9
            with np.errstate(invalid="ignore"):
                M = np.abs(np.asarray(S4[b][0], np.float32) + np.eye(N, dtype=np.float32)) > np.float32(ZERO_TOLERANCE)
        for q in range(P):
            if mode == "KeyQuery":
                Q = X[b] @ W[q, 0].T                           # Q[j] = W x_j
            else:
                a = np.asarray(p["mixer"], np.float64)[q, 0]
                wb = 0.0 if origin else np.asarray(p["weight_bias"], np.float64)[q, 0]
                Wx = X[b] @ W[q, 0].T + wb
                c1, c2 = Wx @ a[:F], Wx @ a[F:]
            for i in range(N):
                nb = np.nonzero(M[i])[0]
                if nb.size == 0:
                    continue
                if mode == "KeyQuery":
                    e = np.array([X[b, i] @ Q[j] for j in nb])
                else:
                    e = c1[nb] + c2[i]
                    e = np.where(e > 0, e, 0.2 * e)
                w = np.exp(e - e.max())
                A[b, q, i, nb] = w / w.sum()
            Z = X[b].copy()
            acc = Z @ Hf[q, :, 0, 0, :].T
            for k in range(1, K):
                Z = A[b, q].T @ Z                              # z_j <- sum_i a_ij z_i
                acc = acc + Z @ Hf[q, :, 0, k, :].T
            Y[b, q] = acc + (0.0 if bias is None else bias)
    if concat:
        out = np.maximum(Y, 0).transpose(0, 2, 1, 3).reshape(B, N, P * F)
    else:
        out = np.maximum(Y.mean(axis=1), 0)
    return out.transpose(0, 2, 1), A[:, :, None]


# ------------------------------------------------------------------------- CNN
def _bn(x, sd, pre, eps=1e-5):
    return tnf.batch_norm(x, sd[pre + ".running_mean"], sd[pre + ".running_var"],
                          sd[pre + ".weight"], sd[pre + ".bias"], False, 0.0, eps)


def _basic_block(x, sd, pre, stride):
    """BasicBlock (resnet_pytorch.py:40-73) with the 1x1-conv+BN downsample."""
    out = tnf.conv2d(x, sd[pre + ".conv1.weight"], None, stride, 1)
    out = torch.relu(_bn(out, sd, pre + ".bn1"))
    out = tnf.conv2d(out, sd[pre + ".conv2.weight"], None, 1, 1)
    out = _bn(out, sd, pre + ".bn2")
    if (pre + ".downsample.0.weight") in sd:
        res = tnf.conv2d(x, sd[pre + ".downsample.0.weight"], None, stride, 0)
        res = _bn(res, sd, pre + ".downsample.1")
    else:
        res = x
    return torch.relu(out + res)


def resnet_forward(x, sd, pre="ConvLayers.0"):
    """ResNet(BasicBlock,[1,1,1]) / ResNetSlim(BasicBlock,[1,1]) eval forward
    (resnet_pytorch.py:495-524 / 403-425); Slim has no layer3."""
    y = tnf.conv2d(x, sd[pre + ".conv1.weight"], None, 1, 1)
    y = torch.relu(_bn(y, sd, pre + ".bn1"))
    y = _basic_block(y, sd, pre + ".layer1.0", 2)
    y = _basic_block(y, sd, pre + ".layer2.0", 1)
    if (pre + ".layer3.0.conv1.weight") in sd:
        y = _basic_block(y, sd, pre + ".layer3.0", 1)
    y = tnf.avg_pool2d(y, 2)
    return tnf.conv2d(y, sd[pre + ".fc.weight"], sd[pre + ".fc.bias"])


def default_cnn_forward(x, sd, pre="ConvLayers"):
    """CNN_mode=Default: 5 x [conv3x3(bias)+BN+ReLU], MaxPool2d(2) after layers 0,2,4
    (decentralplanner_GAT_bottleneck.py:118-147).  Sequential indices: conv, bn, relu
    [, pool] per layer."""
    idx = 0
    for l in range(5):
        x = tnf.conv2d(x, sd["%s.%d.weight" % (pre, idx)], sd["%s.%d.bias" % (pre, idx)], 1, 1)
        x = torch.relu(_bn(x, sd, "%s.%d" % (pre, idx + 1)))
        idx += 3
        if l % 2 == 0:
            x = tnf.max_pool2d(x, 2)
            idx += 1
    return x


def dilated_cnn_forward(x, sd, version=1, pre="ConvLayers"):
    """config.use_dilated of DecentralPlannerNet (graphs/models/decentralplanner.py:57-86: numChannel / numDilated / nPaddingSzie of
    use_dilated_version 1 | 2; :138-162: Conv2d(3 x 3, dilation, padding) + BatchNorm2d + ReLU per layer, MaxPool2d(2, 2) behind
    layers 1 and 3).  Sequential indices: conv, bn, relu [, pool] per layer."""
    dil = [1, 3, 1, 3, 1] if version == 1 else [1, 3, 1, 3]
    idx = 0
    for l, d in enumerate(dil):
        x = tnf.conv2d(x, sd["%s.%d.weight" % (pre, idx)], sd["%s.%d.bias" % (pre, idx)], stride=1, padding=d, dilation=d)
        x = torch.relu(_bn(x, sd, "%s.%d" % (pre, idx + 1)))
        idx += 3
        if l in (1, 3):
            x = tnf.max_pool2d(x, 2, 2)
            idx += 1
    return x


def conv_layers_forward(x, sd, cnn_mode):
    """self.ConvLayers(...) then .view(B*N,-1) (…bottleneck.py:90-147, 294-297);
    Dropout is identity in eval()."""
    if cnn_mode in ("ResNetLarge_withMLP", "ResNetSlim_withMLP"):
        f = resnet_forward(x, sd).flatten(1)
        return tnf.linear(f, sd["ConvLayers.3.weight"], sd["ConvLayers.3.bias"])
    if cnn_mode in ("ResNetLarge", "ResNetSlim"):
        return resnet_forward(x, sd).flatten(1)
    return default_cnn_forward(x, sd).flatten(1)


# ------------------------------------------------------------------ full model
def planner_forward(x, S, sd, cfg, return_parts=False):
    """DecentralPlannerGATNet addGSO + forward in eval mode.
    x (B,N,3,W,H) f32; S (B,N,N) f32|f64 (mutated in place like the reference);
    sd: state_dict (CPU tensors); cfg: object with CNN_mode, attentionMode,
    AttentionConcat, GSO_mode, bottleneckMode, use_dropout.  Returns logits (B*N,5)."""
    B, N = x.shape[0], x.shape[1]
    mode = getattr(cfg, "bottleneckMode", "BottomNeck_only")
    S4 = add_gso(S, cfg.GSO_mode, mode)
    feat = conv_layers_forward(x.reshape(B * N, *x.shape[2:]), sd, cfg.CNN_mode)
    comp = torch.relu(tnf.linear(feat, sd["compressMLP.0.weight"], sd["compressMLP.0.bias"]))
    G = comp.shape[1]
    xg = comp.reshape(B, N, G).permute(0, 2, 1)
    gp = {k: sd["GFL.0." + k] for k in ("mixer", "weight_bias", "filterWeight", "bias", "weight") if "GFL.0." + k in sd}
    yg, aij = gat_layer_forward(xg, S4, gp, cfg.attentionMode, cfg.AttentionConcat)
    shared = yg.permute(0, 2, 1).reshape(B * N, yg.shape[1])
    if mode == "BottomNeck_skipConcat":
        shared = torch.cat((feat, shared), dim=1)
    elif mode == "BottomNeck_skipConcatGNN":
        shared = torch.cat((comp, shared), dim=1)
    elif mode == "BottomNeck_skipAddGNN":
        shared = comp + shared
    h = tnf.linear(shared, sd["actionsMLP.0.weight"], sd["actionsMLP.0.bias"])
    if getattr(cfg, "use_dropout", False):   # Linear-ReLU-Dropout-Linear-Dropout (…:226-236)
        h = tnf.linear(torch.relu(h), sd["actionsMLP.3.weight"], sd["actionsMLP.3.bias"])
    if return_parts:
        return h, dict(feat=feat, comp=comp, gat=shared, aij=aij)
    return h


def planner_gnn_forward(x, S, sd, cfg):
    """DecentralPlannerNet addGSO + forward in eval mode (graphs/models/decentralplanner.py:336-398): encoder, compressMLP,
    ONE GraphFilterBatch (graphML.py:5670-5689 -> BatchLSIGF :5485-5579), ReLU unless cfg.no_ReLU, action MLP.  addGSO of this
    class always scrubs NaN (:346), like the BottomNeck_only GAT file.  S is mutated in place like the reference."""
    B, N = x.shape[0], x.shape[1]
    S4 = add_gso(S, cfg.GSO_mode, "BottomNeck_only")
    if getattr(cfg, "use_dilated", False):      # (decentralplanner.py:138: the dilated CNNs take precedence over CNN_mode)
        feat = dilated_cnn_forward(x.reshape(B * N, *x.shape[2:]), sd, int(getattr(cfg, "use_dilated_version", 1))).flatten(1)
    else:
        feat = conv_layers_forward(x.reshape(B * N, *x.shape[2:]), sd, cfg.CNN_mode)
    comp = torch.relu(tnf.linear(feat, sd["compressMLP.0.weight"], sd["compressMLP.0.bias"]))
    xg = comp.reshape(B, N, comp.shape[1]).permute(0, 2, 1)
    yg = graph_filter_batch_forward(xg, S4, sd["GFL.0.weight"], sd["GFL.0.bias"])
    if not getattr(cfg, "no_ReLU", False):
        yg = torch.relu(yg)
    shared = yg.permute(0, 2, 1).reshape(B * N, yg.shape[1])
    h = tnf.linear(shared, sd["actionsMLP.0.weight"], sd["actionsMLP.0.bias"])
    if getattr(cfg, "use_dropout", False):
        h = tnf.linear(torch.relu(h), sd["actionsMLP.3.weight"], sd["actionsMLP.3.bias"])
    return h


# ------------------------------------------------------------- reference init
def init_state_dict(cfg, seed=1337, perturb_bn=True):
    """Random weights with the reference's shapes and init laws (weights_init
    graphs/weights_initializer.py:11-23; reset_parameters graphML.py:4604-4612), drawn from
    a seeded generator.  Used where no reference-made fixture applies (full-size parity
    and bench).  BN running stats are perturbed so BN folding bugs cannot hide."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def xavier(*shape):
        rf = 1
        for s in shape[2:]:
            rf *= s
        std = math.sqrt(2.0 / (shape[1] * rf + shape[0] * rf))
        return torch.randn(*shape, generator=g) * std

    def bn(pre, c):
        sd[pre + ".weight"] = 1.0 + 0.02 * torch.randn(c, generator=g)
        sd[pre + ".bias"] = (0.1 * torch.randn(c, generator=g)) if perturb_bn else torch.zeros(c)
        sd[pre + ".running_mean"] = (0.2 * torch.randn(c, generator=g)) if perturb_bn else torch.zeros(c)
        sd[pre + ".running_var"] = (0.5 + torch.rand(c, generator=g)) if perturb_bn else torch.ones(c)
        sd[pre + ".num_batches_tracked"] = torch.tensor(0)

    def block(pre, cin, cout):
        sd[pre + ".conv1.weight"] = xavier(cout, cin, 3, 3)
        bn(pre + ".bn1", cout)
        sd[pre + ".conv2.weight"] = xavier(cout, cout, 3, 3)
        bn(pre + ".bn2", cout)
        sd[pre + ".downsample.0.weight"] = xavier(cout, cin, 1, 1)
        bn(pre + ".downsample.1", cout)

    mode = cfg.CNN_mode
    if mode.startswith("ResNet"):
        pre = "ConvLayers.0"
        sd[pre + ".conv1.weight"] = xavier(32, 3, 3, 3)
        bn(pre + ".bn1", 32)
        block(pre + ".layer1.0", 32, 32)
        block(pre + ".layer2.0", 32, 64)
        last = 64
        if "Large" in mode:
            block(pre + ".layer3.0", 64, 128)
            last = 128
        sd[pre + ".fc.weight"] = xavier(128, last, 1, 1)
        sd[pre + ".fc.bias"] = 0.05 * torch.randn(128, generator=g)
        if mode.endswith("_withMLP"):
            sd["ConvLayers.3.weight"] = xavier(cfg.numInputFeatures, 1152)
            sd["ConvLayers.3.bias"] = 0.05 * torch.randn(cfg.numInputFeatures, generator=g)
            nfm = cfg.numInputFeatures
        else:
            nfm = 1152
    else:
        ch = [3, 32, 32, 64, 64, 128]
        idx = 0
        for l in range(5):
            sd["ConvLayers.%d.weight" % idx] = xavier(ch[l + 1], ch[l], 3, 3)
            sd["ConvLayers.%d.bias" % idx] = 0.05 * torch.randn(ch[l + 1], generator=g)
            bn("ConvLayers.%d" % (idx + 1), ch[l + 1])
            idx += 3 + (1 if l % 2 == 0 else 0)
        w = h = cfg.FOV + 2              # decentralplanner_GAT_bottleneck.py:132-140: three MaxPool2d(2), floor mode
        for _ in range(3):
            w, h = (w - 2) // 2 + 1, (h - 2) // 2 + 1
        nfm = 128 * w * h
    bmode = getattr(cfg, "bottleneckMode", "BottomNeck_only")
    G = cfg.bottleneckFeature if bmode in SKIP_MODES else cfg.numInputFeatures
    sd["compressMLP.0.weight"] = xavier(G, nfm)
    sd["compressMLP.0.bias"] = 0.05 * torch.randn(G, generator=g)
    P, K, F = cfg.nAttentionHeads, cfg.nGraphFilterTaps, G
    stdv = 1.0 / math.sqrt(G * P)

    def uni(*shape):
        return (torch.rand(*shape, generator=g) * 2 - 1) * stdv

    sd["GFL.0.mixer"] = uni(P, 1, 2 * F)
    if cfg.attentionMode == "GAT_origin":
        sd["GFL.0.weight"] = uni(P, 1, F, G)
        sd["GFL.0.filterWeight"] = uni(1, K)
        sd["GFL.0.bias"] = uni(F, 1)
    else:
        sd["GFL.0.weight_bias"] = uni(P, 1, F) * (1.0 if perturb_bn else 0.0)
        sd["GFL.0.filterWeight"] = uni(P, F, 1, K, G)
        sd["GFL.0.bias"] = uni(F, 1)
        sd["GFL.0.weight"] = uni(P, 1, G, G) if cfg.attentionMode == "KeyQuery" else uni(P, 1, F, G)
    nin = P * F if cfg.AttentionConcat else F
    if bmode == "BottomNeck_skipConcat":
        nin += nfm
    elif bmode == "BottomNeck_skipConcatGNN":
        nin += G
    if getattr(cfg, "use_dropout", False):
        sd["actionsMLP.0.weight"] = xavier(cfg.numInputFeatures, nin)
        sd["actionsMLP.0.bias"] = torch.zeros(cfg.numInputFeatures)
        sd["actionsMLP.3.weight"] = xavier(5, cfg.numInputFeatures)
        sd["actionsMLP.3.bias"] = torch.zeros(5)
    else:
        sd["actionsMLP.0.weight"] = xavier(5, nin)
        sd["actionsMLP.0.bias"] = 0.05 * torch.randn(5, generator=g)
    return sd
