"""Inference-time parameter folding for the per-agent CNN encoder.

Turns the reference-layout parameters of ConvLayers / compressMLP
(graphs/models/resnet_pytorch.py:40-73, 334-524; decentralplanner_GAT_bottleneck.py:90-166) into
the "encoder pack" consumed by magat_encoder_forward_f32 (include/magat_hip.h):

  off[0]  conv1 weight  [32][27]      (BN folded)          off[1]  conv1 bias [32]
  per BasicBlock l = 0..2 (Slim: 0..1):
  off[2+4l] conv1 weight [Cout][9*Cin]  (ty,tx,c order)    off[3+4l] bias [Cout]
  off[4+4l] [conv2 | downsample] weight [Cout][9*Cout+Cin] off[5+4l] bias_conv2 + bias_down
  off[14] head weight [n_feat][Hp*Wp*Clast]  (fc(+Flatten+Linear) folded, x 1/4 of AvgPool2d(2);
          the GEMM sum-pools the 2x2 windows while loading its A operand)
  off[15] head bias [n_feat]
  off[16] compressMLP weight [G][n_feat]                   off[17] compressMLP bias [G]
  off[18..23] empty since round 5 (the bf16x3 weight planes of round 1's bf16x6 flavour)
  off[24+2l], off[25+2l] the same two weights as f16x2 planes of (w * 2^e) + one float32 2^-e  ("f16x3" GEMM, in_fmt 4)

Every offset is a multiple of 4 floats.  Folding is done in float64 and stored as float32.
"""
import math

import torch

BN_EPS = 1e-5


def _bn_scale_shift(sd, pre):
    s = sd[pre + ".weight"].double() / torch.sqrt(sd[pre + ".running_var"].double() + BN_EPS)
    return s, sd[pre + ".bias"].double() - sd[pre + ".running_mean"].double() * s


def _conv_rows(w, scale):
    """(Cout,Cin,kH,kW) -> [Cout][(ty*kW+tx)*Cin + c], rows scaled."""
    co = w.shape[0]
    return (w.double() * scale.view(-1, 1, 1, 1)).permute(0, 2, 3, 1).reshape(co, -1)


def fold_resnet(sd, H=11, W=11, pre="ConvLayers.0", linear=None, compress=None):
    """sd: state_dict-like {name: tensor}.  linear = (weight, bias) of ConvLayers.3 for *_withMLP modes
    else None.  compress = (weight, bias) of compressMLP.0 or None.
    Returns (pack float32 1-D CPU tensor, offsets list[32], meta dict)."""
    sd = {k: v.detach().cpu() for k, v in sd.items() if k.startswith(pre)}
    parts, offs = [], [0] * 32
    cursor = [0]

    def put(slot, t):
        t = t.reshape(-1)
        offs[slot] = cursor[0]
        pad = (-t.numel()) % 4
        if pad:
            t = torch.cat((t, torch.zeros(pad, dtype=t.dtype)))
        parts.append(t)
        cursor[0] += t.numel()

    s, b = _bn_scale_shift(sd, pre + ".bn1")
    put(0, sd[pre + ".conv1.weight"].double().reshape(32, 27) * s.view(-1, 1))
    put(1, b)
    large = (pre + ".layer3.0.conv1.weight") in sd
    nblocks = 3 if large else 2
    clast = 32
    for l in range(nblocks):
        bp = "%s.layer%d.0" % (pre, l + 1)
        w1 = sd[bp + ".conv1.weight"]
        cout, cin = w1.shape[0], w1.shape[1]
        s1, b1 = _bn_scale_shift(sd, bp + ".bn1")
        put(2 + 4 * l, _conv_rows(w1, s1))
        put(3 + 4 * l, b1)
        s2, b2 = _bn_scale_shift(sd, bp + ".bn2")
        w2 = _conv_rows(sd[bp + ".conv2.weight"], s2)
        if (bp + ".downsample.0.weight") in sd:
            sdn, bdn = _bn_scale_shift(sd, bp + ".downsample.1")
            wd = sd[bp + ".downsample.0.weight"].double().reshape(cout, cin) * sdn.view(-1, 1)
        else:
            wd, bdn = torch.eye(cout, cin, dtype=torch.float64), torch.zeros(cout, dtype=torch.float64)
        put(4 + 4 * l, torch.cat((w2, wd), dim=1))
        put(5 + 4 * l, b2 + bdn)
        clast = cout
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    Hp, Wp = Ho // 2, Wo // 2
    fc = sd[pre + ".fc.weight"].double().reshape(-1, clast)       # (128, clast)
    fcb = sd[pre + ".fc.bias"].double()
    nq = fc.shape[0]
    if linear is not None:
        Lw, Lb = linear[0].detach().cpu().double(), linear[1].detach().cpu().double()
        assert Lw.shape[1] == nq * Hp * Wp, "Linear in_features must equal 128*pooled pixels"
        L4 = Lw.reshape(-1, nq, Hp, Wp)
        W2 = torch.einsum("oqyx,qc->oyxc", L4, fc)                 # (n_feat, Hp, Wp, clast)
        b2 = Lb + torch.einsum("oqyx,q->o", L4, fcb)
        n_feat = Lw.shape[0]
    else:
        eye = torch.eye(Hp * Wp, dtype=torch.float64).reshape(Hp, Wp, Hp, Wp)
        W2 = torch.einsum("qc,abyx->qabyxc", fc, eye).reshape(nq * Hp * Wp, Hp, Wp, clast)
        b2 = fcb.repeat_interleave(Hp * Wp)
        n_feat = nq * Hp * Wp
    put(14, 0.25 * W2)      # the kernel sum-pools 2x2 on load; the 1/4 lives here
    put(15, b2)
    n_comp = 0
    if compress is not None:
        put(16, compress[0].detach().cpu().double())
        put(17, compress[1].detach().cpu().double())
        n_comp = compress[0].shape[0]
    pack = torch.cat(parts).to(torch.float32).contiguous()
    if True:
        # block-conv weights once more for the split-MFMA kernels (csrc/conv_gemm_bf16x6.hip).  Slots 18..23 (the bf16x3
        # planes of round 1's bf16x6 flavour) stay empty since round 5.
        raws = []
        cursor_f = pack.numel()
        n_f32 = pack.numel()
        pairs = []
        for l in range(nblocks):
            pairs += [(18 + 2 * l, 2 + 4 * l), (19 + 2 * l, 4 + 4 * l)]
        # As f16x2 planes of the power-of-two-scaled weights followed by the inverse scale ("f16x3" GEMM, in_fmt 4):
        # off[24+2l] = layer(l+1).conv1, off[25+2l] = [conv2|downsample]
        # Each f16 block is followed by a second copy with the K columns of every 32-wide slab permuted for activations
        # stored as f16 plane granules (magat_hip.h in_gl = 2; offs[30] != 0 marks their presence): the producer's MFMA
        # lane holds channels 4h + 8g + c (g, c = 0..3) of a 32-channel tile, and writes quads g = 2ks, 2ks+1 as ONE
        # 16-byte operand, so operand slot 16ks + 8h + i carries channel 16ks + 8(i>>2) + 4h + (i&3).
        perm32 = torch.tensor([16 * (q >> 4) + 8 * ((q & 7) >> 2) + 4 * ((q >> 3) & 1) + (q & 3) for q in range(32)])
        couts = [32, 32, 64, 64, 128, 128]
        for (slot, src), cout in zip(pairs, couts):
            nxt = sorted(o for o in offs[:18] if o > offs[src])
            end = nxt[0] if nxt else n_f32
            w32 = pack[offs[src]:end]
            blk, _ = split_f16x2(w32)
            pad = (-blk.numel()) % 4
            if pad:
                blk = torch.cat((blk, torch.zeros(pad)))
            offs[slot + 6] = cursor_f
            cursor_f += blk.numel()
            raws.append(blk)
            w2d = w32.reshape(cout, -1)
            assert w2d.shape[1] % 32 == 0
            idx = (torch.arange(w2d.shape[1]) // 32) * 32
            idx = idx + perm32.repeat(w2d.shape[1] // 32)
            blk2, _ = split_f16x2(w2d[:, idx].reshape(-1))
            if pad:
                blk2 = torch.cat((blk2, torch.zeros(pad)))
            assert blk2.numel() == blk.numel()
            cursor_f += blk2.numel()
            raws.append(blk2)
            if slot == pairs[0][0]:
                offs[30] = offs[slot + 6] + blk.numel()        # first permuted copy (non-zero = copies present)
        pack = torch.cat([pack] + raws).contiguous()
    # BasicBlock chain kernel (csrc/block_fused.hip; 6x6 maps only, i.e. the reference's 11x11 input): layer1.conv2+ds,
    # layer2.conv1, layer2.conv2+ds as fragment-major f16 planes, appended to the pack; its float offset goes to meta["chain"]
    chain_off = 0
    if Ho == 6 and Wo == 6 and nblocks >= 2:
        def rows(slot, cout):
            nxt = sorted(o for o in offs[:18] if o > offs[slot])
            return pack[offs[slot]:(nxt[0] if nxt else n_f32)].reshape(cout, -1)
        blocks = [pack_chain_weights(rows(4, 32)[:, :9 * 32 + 32], 32, 32),
                  pack_chain_weights(rows(6, 64)[:, :9 * 32], 32, 0),
                  pack_chain_weights(rows(8, 64)[:, :9 * 64 + 32], 64, 32)]
        chain_off = pack.numel()
        pack = torch.cat([pack] + blocks).contiguous()
    chain3_off = 0
    if chain_off and large:
        def rows3(slot):
            nxt = sorted(o for o in offs[:18] if o > offs[slot])
            return pack[offs[slot]:(nxt[0] if nxt else n_f32)].reshape(128, -1)
        chain3_off = pack.numel()
        pack = torch.cat([pack] + pack_block3_weights(rows3(10)[:, :9 * 64], rows3(12)[:, :9 * 128 + 64])).contiguous()
    # the head once more as f16x2 planes (in_fmt 4): with the layer3 kernel's pooled output the head is a plain valid conv
    # over the pooled map, 5x faster on the split-MFMA kernel than on the float32 matrix cores
    nxt = sorted(o for o in offs[:18] if o > offs[14])
    head16, _ = split_f16x2(pack[offs[14]:(nxt[0] if nxt else n_f32)])
    pad = (-head16.numel()) % 4
    if pad:
        head16 = torch.cat((head16, torch.zeros(pad)))
    head16_off = pack.numel()
    pack = torch.cat([pack, head16]).contiguous()
    # ... and compressMLP's weight the same way (it follows the head on the same kernel)
    comp16_off = 0
    if n_comp > 0:
        comp16, _ = split_f16x2(pack[offs[16]:offs[16] + n_comp * n_feat])
        pad = (-comp16.numel()) % 4
        if pad:
            comp16 = torch.cat((comp16, torch.zeros(pad)))
        comp16_off = pack.numel()
        pack = torch.cat([pack, comp16]).contiguous()
    # the head and compressMLP once more FRAGMENT-major (ABI 8; csrc/block_lat.hip: the one-agent-per-workgroup encoder of the
    # batch-1 step runs both layers in the chain kernel's epilogue): the same f16 planes and scale as head16 / comp16, in the
    # order a wave fetches them - no weight slab through LDS there, 1 KB blocks straight into registers
    headfrag_off = compfrag_off = 0
    if chain3_off and n_comp in (32, 64, 128) and n_feat == 128 and clast == 128 and Hp * Wp == 9:
        nxt = sorted(o for o in offs[:18] if o > offs[14])
        hrows = pack[offs[14]:(nxt[0] if nxt else n_f32)].reshape(n_feat, Hp * Wp * clast)
        # k steps in the long-K head's order (conv_gemm_bf16x6.hip direct kernel, korder 1): 32-channel slab outer, pooled cell
        # inner, two 16-channel k steps per (slab, cell)
        hsteps = [cell * clast + 32 * cs + 16 * ks for cs in range(clast // 32) for cell in range(Hp * Wp) for ks in range(2)]
        headfrag_off = pack.numel()
        pack = torch.cat([pack, pack_frag_natural(hrows, hsteps)]).contiguous()
        crows = pack[offs[16]:offs[16] + n_comp * n_feat].reshape(n_comp, n_feat)
        compfrag_off = pack.numel()
        pack = torch.cat([pack, pack_frag_natural(crows, [16 * ks for ks in range(n_feat // 16)])]).contiguous()
    # layer1.conv1 fragment-major for the eight-agent-group stem kernel (csrc/block_fused.hip stem8_kernel; 11x11 maps)
    l1frag_off = 0
    if H == 11 and W == 11:
        nxt = sorted(o for o in offs[:18] if o > offs[2])
        w1rows = pack[offs[2]:(nxt[0] if nxt else n_f32)][:32 * 288].reshape(32, 288)
        l1frag_off = pack.numel()
        pack = torch.cat([pack, pack_chain_weights(w1rows, 32, 0)]).contiguous()
    meta = dict(variant=0 if large else 1, H=H, W=W, n_feat=n_feat, n_comp=n_comp, clast=clast, chain=chain_off,
                chain3=chain3_off, head16=head16_off, comp16=comp16_off, l1frag=l1frag_off, headfrag=headfrag_off,
                compfrag=compfrag_off)
    return pack, offs, meta


SCALED_BLOCK_FLOATS = 1352      # magat_hip.h magat_encoder_desc.scaled_off


def scale_exponent(m, target_log2=10, lo=-14, hi=24):
    """Power-of-two exponent e such that m * 2^e lands at ~2^target_log2 (m: largest |activation| of a layer; 0 -> 0)."""
    if not (m > 0.0) or not math.isfinite(m):
        return 0
    return max(lo, min(hi, target_log2 - int(math.floor(math.log2(m))) - 1))


def fold_activation_scales(pack, offs, meta, absmax):
    """The ACTIVATION-SCALE block of the fused encoder path (include/magat_hip.h "Activation scales").  absmax: the 16 floats of
    magat_encoder_calibrate_f32.  Every BasicBlock keeps ONE scale for its input, its conv1 output and the residual input of
    its conv2 (they meet in one accumulator); scales change in the epilogue of conv2: s1 (stem .. layer1.conv1), s2
    (layer1 out .. layer2.conv1), s3 (layer2 out .. layer3.conv1); the pooled layer3 output, feat and comp stay at their
    true scale in HBM (other kernels read them) and are scaled by their float32 loaders (head_in, feat_in).
    Returns (block float32 [SCALED_BLOCK_FLOATS], dict of the exponents)."""
    a = [float(v) for v in absmax]
    s1 = scale_exponent(max(a[0], a[1]))
    # the stem's INPUT is not scaled (binary state maps): its scale rides in the stem weights, which the fused stem kernel
    # splits into f16 planes of 16 w - keep 16 w 2^s1 inside the planes' range whatever the calibration batch looked like
    w0max = float(max(pack[offs[0]:offs[0] + 864].abs().max(), pack[offs[1]:offs[1] + 32].abs().max()))
    if w0max > 0.0:
        s1 = min(s1, 15 - 4 - int(math.ceil(math.log2(w0max))))
    s2 = scale_exponent(max(a[2], a[3]))
    s3 = scale_exponent(max(a[4], a[5]))
    e_head = scale_exponent(4.0 * a[6])          # (the head reads 2x2 SUMS of layer3's output)
    e_feat = scale_exponent(a[7])
    f = lambda e: 2.0 ** e

    def seg(slot, n):
        return pack[offs[slot]:offs[slot] + n].double()
    blk = torch.zeros(SCALED_BLOCK_FLOATS, dtype=torch.float64)
    blk[0:864] = seg(0, 864) * f(s1)
    blk[864:896] = seg(1, 32) * f(s1)
    blk[896:928] = seg(3, 32) * f(s1)            # layer1.conv1 bias (its 1 / weight-scale is unchanged: in and out share s1)
    blk[928:960] = seg(5, 32) * f(s2)            # stage A  (layer1.conv2 + downsample): s1 -> s2
    blk[960:1024] = seg(7, 64) * f(s2)           # stage B  (layer2.conv1)
    blk[1024:1088] = seg(9, 64) * f(s3)          # stage C  (layer2.conv2 + downsample): s2 -> s3
    blk[1088:1216] = seg(11, 128) * f(s3)        # layer3.conv1
    blk[1216:1344] = seg(13, 128)                # layer3.conv2 + downsample: s3 -> true scale
    ch, c3 = meta["chain"], meta["chain3"]
    nA, nB, nC = 10240, 18432, 38912             # floats of the three chain weight blocks (block_fused.hip chain_block_bytes)
    n1, n2a, n2b = 73728, 73728, 81920
    sA, sB, sC = float(pack[ch + nA]), float(pack[ch + nA + 4 + nB]), float(pack[ch + nA + 4 + nB + 4 + nC])
    s31, s32 = float(pack[c3 + n1]), float(pack[c3 + n1 + 4 + n2a + 4 + n2b])
    blk[1344] = sA * f(s2 - s1)
    blk[1345] = sB
    blk[1346] = sC * f(s3 - s2)
    blk[1347] = s31
    blk[1348] = s32 * f(-s3)
    blk[1349] = f(e_head)
    blk[1350] = f(e_feat)
    return blk.to(torch.float32), dict(s1=s1, s2=s2, s3=s3, head_in=e_head, feat_in=e_feat)


def pack_block3_weights(w1, w2):
    """layer3 kernel (csrc/block_fused.hip block3_kernel): conv1 rows [128][9*64] and [conv2 | downsample] rows
    [128][9*128 + 64] -> three fragment-major blocks: conv1; conv2 over intermediate channels 0..63; conv2 over channels
    64..127 followed by the residual columns.  The two conv2 blocks share one power-of-two scale (they feed the same
    accumulators)."""
    w2 = w2.detach().float().cpu()
    taps = w2[:, :9 * 128].reshape(128, 9, 128)
    lo = taps[:, :, :64].reshape(128, 9 * 64)
    hi = torch.cat((taps[:, :, 64:].reshape(128, 9 * 64), w2[:, 9 * 128:]), dim=1)
    mx = float(w2.abs().max())
    e = 0 if mx == 0.0 else 13 - int(math.floor(math.log2(mx)))
    e = max(-14, min(e, 24))
    return [pack_chain_weights(w1, 64, 0), pack_chain_weights(lo, 64, 0, e=e), pack_chain_weights(hi, 64, 64, e=e)]


def chain_channel(ks, fh, i):
    """channel carried by element i of MFMA lane half fh in k step ks of a plane-granule map (magat_hip.h in_gl = 2)."""
    return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (i >> 2) + 4 * fh + (i & 3)


def pack_chain_weights(w, cin, c2, e=None):
    """[Cout][9*cin + c2] float32 BN-folded rows ((ty, tx, c) order, then the 1x1 residual columns) -> the fragment-major
    operand block of csrc/block_fused.hip: for channel tile ct, tap u = 0..8 (9 = residual segment), k step ks, plane pl
    one 1 KB block [64 lanes][8 halves]: lane l holds, for output channel 32 ct + (l & 31), the weights of the channels
    chain_channel(ks, l >> 5, i), i = 0..7, of that tap - the order in which the activation lanes hold their operand.
    Values are f16 planes of w * 2^e (split_f16x2's scale); 4 floats [2^-e, 0, 0, 0] follow."""
    w = w.detach().float().cpu()
    cout = w.shape[0]
    assert w.shape[1] == 9 * cin + c2 and cout % 32 == 0 and cin % 16 == 0 and c2 % 16 == 0
    if e is None:
        mx = float(w.abs().max())
        e = 0 if mx == 0.0 else 13 - int(math.floor(math.log2(mx)))
        e = max(-14, min(e, 24))
    ws = w * (2.0 ** e)
    h1 = ws.half()
    h2 = (ws - h1.float()).half()
    lane = torch.arange(64)
    n_in, fh = lane & 31, lane >> 5
    i8 = torch.arange(8)
    out = []
    for ct in range(cout // 32):
        segs = [(u, cin // 16, u * cin) for u in range(9)] + ([(9, c2 // 16, 9 * cin)] if c2 else [])
        for u, nks, base in segs:
            for ks in range(nks):
                ch = (32 * (ks >> 1) + 16 * (ks & 1) + 8 * (i8.view(1, 8) >> 2) + 4 * fh.view(64, 1) + (i8.view(1, 8) & 3))
                col = base + ch                                          # [64][8]
                row = (32 * ct + n_in).view(64, 1).expand(64, 8)
                for plane in (h1, h2):
                    out.append(plane[row, col].reshape(-1))
    blk = torch.cat(out).view(torch.int16)
    return torch.cat((blk.view(torch.float32), torch.tensor([2.0 ** (-e), 0.0, 0.0, 0.0], dtype=torch.float32)))


def pack_frag_natural(w, steps):
    """[Cout][K] float32 rows -> fragment-major f16 planes with the k values in NATURAL order: for channel tile ct and k step s
    (steps[s] = first column of its 16) one 1 KB block per plane [64 lanes][8 halves] - lane l holds, for output channel
    32 ct + (l & 31), the weights of columns steps[s] + 8 (l >> 5) + i, i = 0..7: the A operand of v_mfma_f32_32x32x16_f16
    against an activation operand formed from float32 values in channel order (the long-K head's and compressMLP's loaders,
    conv_gemm_bf16x6.hip).  Planes and scale exactly as split_f16x2 makes them (h1 = f16(w 2^e), h2 = f16(w 2^e - h1)); 4 floats
    [2^-e, 0, 0, 0] follow."""
    w = w.detach().float().cpu()
    cout = w.shape[0]
    assert cout % 32 == 0
    mx = float(w.abs().max())
    e = 0 if mx == 0.0 else 13 - int(math.floor(math.log2(mx)))
    e = max(-14, min(e, 24))
    ws = w * (2.0 ** e)
    h1 = ws.half()
    h2 = (ws - h1.float()).half()
    lane = torch.arange(64)
    st = torch.tensor(steps, dtype=torch.long)
    col = st.view(-1, 1, 1) + 8 * (lane >> 5).view(1, 64, 1) + torch.arange(8).view(1, 1, 8)      # [step][lane][8]
    out = []
    for ct in range(cout // 32):
        row = (32 * ct + (lane & 31)).view(1, 64, 1).expand(len(steps), 64, 8)
        out.append(torch.stack((h1[row, col], h2[row, col]), dim=1).reshape(-1))                    # [step][plane][lane][8]
    blk = torch.cat(out).view(torch.int16)
    return torch.cat((blk.view(torch.float32), torch.tensor([2.0 ** (-e), 0.0, 0.0, 0.0], dtype=torch.float32)))


def split_bf16x3(t):
    """fp32 tensor -> (3, *t.shape) bfloat16 planes with t == p0 + p1 + p2 to 2^-24 (round-to-nearest-even each)."""
    t = t.float()
    p0 = t.bfloat16()
    r = t - p0.float()
    p1 = r.bfloat16()
    p2 = (r - p1.float()).bfloat16()
    return torch.stack((p0, p1, p2), dim=0).contiguous()


def split_f16x2(t):
    """fp32 weight tensor -> (float32 view of [2 f16 planes of t * 2^e | one float32 2^-e], e): the operand format of the
    "f16x3" GEMM (csrc/conv_gemm_bf16x6.hip, in_fmt 4).  2^e scales the largest |weight| into [2^13, 2^14) so that both
    planes (h1 = f16(w 2^e), h2 = f16(w 2^e - h1)) are normal half-precision numbers; h1 + h2 carries 22 significand bits."""
    t = t.detach().float().cpu().reshape(-1)
    mx = float(t.abs().max())
    e = 0 if mx == 0.0 else 13 - int(math.floor(math.log2(mx)))
    e = max(-14, min(e, 24))
    ts = t * (2.0 ** e)
    h1 = ts.half()
    h2 = (ts - h1.float()).half()
    planes = torch.cat((h1, h2)).view(torch.int16)
    if planes.numel() % 2:
        planes = torch.cat((planes, torch.zeros(1, dtype=torch.int16)))
    out = torch.cat((planes.view(torch.float32), torch.tensor([2.0 ** (-e)], dtype=torch.float32)))
    return out, e


def default_cnn_cells(H, W):
    """Map size after the three MaxPool2d(2) of CNN_mode "Default" (floor mode: w -> w // 2;
    decentralplanner_GAT_bottleneck.py:118-147)."""
    for _ in range(3):
        H, W = H // 2, W // 2
    return H, W


def cell_major_columns(w, hf, wf, c=128):
    """Weight whose columns index the reference's Flatten of a (c, hf, wf) map -> the same weight over features stored
    (cell, channel)-ordered, the layout the HIP encoder writes the Default CNN's features in."""
    lead = w.shape[0]
    return w.reshape(lead, c, hf, wf).permute(0, 2, 3, 1).reshape(lead, hf * wf * c)


def fold_default_cnn(sd, H=11, W=11, pre="ConvLayers", compress=None):
    """CNN_mode "Default" (decentralplanner_GAT_bottleneck.py:118-147): 5 x [Conv2d(bias) + BatchNorm2d + ReLU] with
    MaxPool2d(2) after layers 0, 2, 4.  Sequential indices: conv at 0, 4, 7, 11, 14 (bn = conv+1).  Pack (variant 2):
    off[0..1] conv0 [32][27] / bias, off[2+2i], off[3+2i] conv i+1 weight [Cout][9*Cin] / bias, off[14] identity
    [128][128] (the last max-pool runs as a pooled 1x1 GEMM), off[16..17] compressMLP.  Any map size whose last pool
    leaves at least one cell (H, W >= 8): the features are 128 * hf * wf wide, (cell, channel)-ordered in the kernels'
    buffers, and the compressMLP columns are permuted here to read them (meta["cells"] = (hf, wf) for other readers)."""
    sd = {k: v.detach().cpu() for k, v in sd.items() if k.startswith(pre)}
    parts, offs, cursor = [], [0] * 32, [0]
    hf, wf = default_cnn_cells(H, W)
    if hf < 1 or wf < 1:
        raise ValueError("CNN_mode Default needs maps of at least 8 x 8 (three MaxPool2d(2)); got %d x %d" % (H, W))

    def put(slot, t):
        t = t.reshape(-1).double()
        offs[slot] = cursor[0]
        pad = (-t.numel()) % 4
        if pad:
            t = torch.cat((t, torch.zeros(pad, dtype=t.dtype)))
        parts.append(t)
        cursor[0] += t.numel()

    idx, convs = 0, []
    for l in range(5):
        convs.append(idx)
        idx += 3 + (1 if l % 2 == 0 else 0)
    for l, ci in enumerate(convs):
        w, b = sd["%s.%d.weight" % (pre, ci)].double(), sd["%s.%d.bias" % (pre, ci)].double()
        s, sh = _bn_scale_shift(sd, "%s.%d" % (pre, ci + 1))
        bias = b * s + sh
        if l == 0:
            put(0, w.reshape(32, 27) * s.view(-1, 1))
            put(1, bias)
        else:
            put(2 + 2 * (l - 1), _conv_rows(w, s))
            put(3 + 2 * (l - 1), bias)
    put(14, torch.eye(128, dtype=torch.float64))
    n_comp = 0
    n_feat = 128 * hf * wf
    if compress is not None:
        cw = compress[0].detach().cpu().double()
        assert cw.shape[1] == n_feat, "compressMLP in_features must equal 128 * pooled cells"
        put(16, cell_major_columns(cw, hf, wf))
        put(17, compress[1].detach().cpu().double())
        n_comp = compress[0].shape[0]
    pack = torch.cat(parts).to(torch.float32).contiguous()
    return pack, offs, dict(variant=2, H=H, W=W, n_feat=n_feat, n_comp=n_comp, clast=128, cells=(hf, wf))


DILATED_CNN = {1: dict(chans=[3, 32, 32, 64, 64, 128], dil=[1, 3, 1, 3, 1], variant=3),
               2: dict(chans=[3, 32, 32, 64, 64], dil=[1, 3, 1, 3], variant=4)}


def dilated_cnn_cells(H, W):
    """Map size behind the dilated CNNs of DecentralPlannerNet (decentralplanner.py:138-162): every convolution keeps the size
    (padding = dilation), MaxPool2d(2) behind layers 1 and 3."""
    return (H // 2) // 2, (W // 2) // 2


def fold_dilated_cnn(sd, version, H=11, W=11, pre="ConvLayers", compress=None):
    """config.use_dilated of DecentralPlannerNet (graphs/models/decentralplanner.py:57-86, 138-162): version 1 = 5 x, version 2 =
    4 x [Conv2d(3 x 3, dilation = padding = 1 3 1 3 (1), bias) + BatchNorm2d + ReLU] with MaxPool2d(2) behind layers 1 and 3.
    Sequential indices: conv at 0, 3, 7, 10 (, 14).  Pack as fold_default_cnn's (encoder variant 3 / 4): off[0..1] conv0, off[2+2i],
    off[3+2i] conv i+1, off[14] identity [clast][clast] (version 2 ends on a pool), off[16..17] compressMLP with its columns
    permuted from the reference's (channel, y, x) Flatten to the kernels' (cell, channel) order."""
    spec = DILATED_CNN[version]
    chans = spec["chans"]
    nl = len(chans) - 1
    sd = {k: v.detach().cpu() for k, v in sd.items() if k.startswith(pre)}
    parts, offs, cursor = [], [0] * 32, [0]
    hf, wf = dilated_cnn_cells(H, W)
    if hf < 1 or wf < 1:
        raise ValueError("the dilated CNNs need maps of at least 4 x 4; got %d x %d" % (H, W))

    def put(slot, t):
        t = t.reshape(-1).double()
        offs[slot] = cursor[0]
        pad = (-t.numel()) % 4
        if pad:
            t = torch.cat((t, torch.zeros(pad, dtype=t.dtype)))
        parts.append(t)
        cursor[0] += t.numel()

    idx = 0
    for l in range(nl):
        w, b = sd["%s.%d.weight" % (pre, idx)].double(), sd["%s.%d.bias" % (pre, idx)].double()
        s_, sh = _bn_scale_shift(sd, "%s.%d" % (pre, idx + 1))
        bias = b * s_ + sh
        if l == 0:
            put(0, w.reshape(32, 27) * s_.view(-1, 1))
            put(1, bias)
        else:
            put(2 + 2 * (l - 1), _conv_rows(w, s_))
            put(3 + 2 * (l - 1), bias)
        idx += 3 + (1 if l in (1, 3) else 0)
    clast = chans[-1]
    put(14, torch.eye(clast, dtype=torch.float64))
    n_feat, n_comp = clast * hf * wf, 0
    if compress is not None:
        cw = compress[0].detach().cpu().double()
        assert cw.shape[1] == n_feat, "compressMLP in_features must equal clast * pooled cells"
        put(16, cell_major_columns(cw, hf, wf, c=clast))
        put(17, compress[1].detach().cpu().double())
        n_comp = compress[0].shape[0]
    pack = torch.cat(parts).to(torch.float32).contiguous()
    return pack, offs, dict(variant=spec["variant"], H=H, W=W, n_feat=n_feat, n_comp=n_comp, clast=clast, cells=(hf, wf))

