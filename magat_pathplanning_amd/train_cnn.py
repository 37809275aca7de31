"""Training of the per-agent ResNet encoders on the HIP kernels (round 4).

The reference trains the whole module through autograd (agents/decentralplannerlocal_OnlineExpert_GAT.py:556-567); the
convolutions of graphs/models/resnet_pytorch.py:40-73, 427-524 are 96 % of a training step's FLOPs.  Here every convolution of
the trunk - training-mode forward, input gradient and weight gradient - runs on this library's float32 matrix-core kernels over
PIXEL-MAJOR activations [pixel][agent][channel] (the layout magat_conv_gemm_f32 is built around):

  forward          magat_conv_gemm_f32 (bias = NULL, relu = 0)
  input gradient   magat_conv_gemm_f32 again: dX = conv(dY, mirrored taps, channel roles swapped); a strided convolution
                   takes the zero-stuffed dY (the stride-2 3x3 of layer1 and its 1x1 downsample)
  weight gradient  magat_conv_wgrad_f32 (csrc/conv_train.hip: the contraction over (pixel, agent) rows on
                   v_mfma_f32_32x32x2_f32, partial sums per agent chunk added in a fixed order - deterministic)

  BatchNorm (+ReLU) magat_bn_train_{forward,backward}_f32: batch statistics in one streaming pass, running-stat updates exactly as
                   torch.nn.BatchNorm2d makes them, ReLU and its mask fused

The residual adds and the 2 x 2 average / max pools are torch ops on the GPU over the same pixel-major tensors.  CNN_mode
'Default' (conv + BN + ReLU + MaxPool stacks) runs on the same kernels (conv_stack_forward).
"""
import ctypes

import torch
import torch.nn.functional as tnf

from . import _native as nat


def _conv_gemm(x, wt, hin, win, kh, kw, stride, pad, hout, wout):
    """x [hin*win][M][Cin] float32 contiguous, wt [Cout][kh*kw*Cin] -> [hout*wout][M][Cout] (no bias, no activation)."""
    lib = nat.lib()
    _, M, cin = x.shape
    cout = wt.shape[0]
    dev = x.device
    out = torch.empty(hout * wout, M, cout, dtype=torch.float32, device=dev)
    d = nat.ConvGemmDesc()
    d.inp, d.wt, d.bias, d.out = x.data_ptr(), wt.data_ptr(), None, out.data_ptr()
    d.in_pix_stride, d.out_pix_stride = M * cin, M * cout
    d.M, d.Cin, d.lda = M, cin, cin
    d.Hin, d.Win, d.kH, d.kW, d.stride, d.pad, d.Hout, d.Wout = hin, win, kh, kw, stride, pad, hout, wout
    d.Cout, d.ldc, d.relu = cout, cout, 0
    with torch.cuda.device(dev):
        nat.check(lib.magat_conv_gemm_f32(ctypes.byref(d), nat.current_stream(dev)), "magat_conv_gemm_f32 (training)")
    return out


def _pad4(t):
    """channel count (last axis) up to a multiple of 4 with zeros (the kernels' row strides)"""
    c = t.shape[-1]
    return t if c % 4 == 0 else tnf.pad(t, (0, 4 - c % 4))


class _ConvPixelMajor(torch.autograd.Function):
    """y = conv2d(x, weight, stride, pad) over pixel-major tensors: x [hin*win][M][Cin4], y [hout*wout][M][Cout]; weight in
    torch's (Cout, Cin, kh, kw) layout (Cin4 = Cin rounded up to 4: the padded channels are zeros)."""

    @staticmethod
    def forward(ctx, x, weight, hin, win, stride, pad):
        cout, cin, kh, kw = weight.shape
        hout, wout = (hin + 2 * pad - kh) // stride + 1, (win + 2 * pad - kw) // stride + 1
        x = x.contiguous()
        w = _pad4(weight.detach().float().permute(0, 2, 3, 1))                    # (Cout, kh, kw, Cin4)
        assert w.shape[3] == x.shape[2], (w.shape, x.shape)
        y = _conv_gemm(x, w.reshape(cout, -1).contiguous(), hin, win, kh, kw, stride, pad, hout, wout)
        ctx.save_for_backward(x, weight)
        ctx.geom = (hin, win, stride, pad, hout, wout)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        hin, win, stride, pad, hout, wout = ctx.geom
        cout, cin, kh, kw = weight.shape
        _, M, cin4 = x.shape
        dev = x.device
        dy = dy.contiguous().float()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            # dX[i] = sum_taps W[tap]^T dY[(i + pad - tap) / stride]: the stride-1 convolution of the zero-stuffed dY with
            # the mirrored taps, pad' = k - 1 - pad; rows of X the forward never reached get no tap and stay zero
            if stride == 1:
                dyz, hz, wz = dy, hout, wout
            else:
                hz, wz = (hout - 1) * stride + 1, (wout - 1) * stride + 1
                dyz = torch.zeros(hz, wz, M, cout, dtype=torch.float32, device=dev)
                dyz[::stride, ::stride] = dy.view(hout, wout, M, cout)
                dyz = dyz.view(hz * wz, M, cout)
            wr = weight.detach().float().permute(1, 2, 3, 0).flip(1, 2)                 # [Cin][u][v][co], taps mirrored: one copy
            if cin4 != cin:
                wr = tnf.pad(wr, (0, 0, 0, 0, 0, 0, 0, cin4 - cin))                      # (zero rows for the padded channels)
            wr = wr.reshape(cin4, kh * kw * cout).contiguous()        # (flip keeps the permuted strides: this is the copy that counts)
            dx = _conv_gemm(dyz, wr, hz, wz, kh, kw, 1, kh - 1 - pad, hin, win)
        if ctx.needs_input_grad[1]:
            lib = nat.lib()
            nfl = lib.magat_conv_wgrad_workspace_floats(M, cin4, cin, cout, kh, kw, hout * wout)
            part = torch.empty(nfl, dtype=torch.float32, device=dev)
            chunks = ctypes.c_int(0)
            with torch.cuda.device(dev):
                nat.check(lib.magat_conv_wgrad_f32(nat.ptr(x), M * cin4, cin4, nat.ptr(dy), M * cout, cout, nat.ptr(part),
                                                   ctypes.byref(chunks), M, cin4, cin, cout, hin, win, hout, wout, kh, kw,
                                                   stride, pad, nat.current_stream(dev)), "magat_conv_wgrad_f32")
            n = chunks.value                          # partial sums per agent chunk, each in the weight's own layout
            dw = part[:n * weight.numel()].view(n, cout, cin, kh, kw).sum(dim=0).to(weight.dtype)
        return dx, dw, None, None, None, None


class _BatchNormTrain(torch.autograd.Function):
    """Training-mode BatchNorm (+ ReLU) over rows [R][C] on magat_bn_train_{forward,backward}_f32 (csrc/conv_train.hip)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, factor, eps, relu):
        lib = nat.lib()
        R, C = x.shape
        dev = x.device
        x = x.contiguous()
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd = torch.empty(C, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.magat_bn_train_workspace_floats(R, C), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            nat.check(lib.magat_bn_train_forward_f32(nat.ptr(x), nat.ptr(y), R, C, nat.ptr(gamma.detach()), nat.ptr(beta.detach()),
                                                     nat.ptr(running_mean), nat.ptr(running_var), float(factor), float(eps),
                                                     int(relu), nat.ptr(mean), nat.ptr(invstd), nat.ptr(ws),
                                                     nat.current_stream(dev)), "magat_bn_train_forward_f32")
        # the kernel updated the running statistics in place: tell torch (version counters are what the inference path's
        # weights key - planner._weights_key - and autograd's in-place checks look at)
        for buf in (running_mean, running_var):
            if buf is not None:
                torch.autograd.graph.increment_version(buf)
        ctx.save_for_backward(x, y, gamma, mean, invstd)
        ctx.relu = bool(relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = nat.lib()
        x, y, gamma, mean, invstd = ctx.saved_tensors
        R, C = x.shape
        dev = x.device
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.magat_bn_train_workspace_floats(R, C), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            nat.check(lib.magat_bn_train_backward_f32(nat.ptr(x), nat.ptr(y), nat.ptr(dy), nat.ptr(dx), R, C, nat.ptr(gamma.detach()),
                                                      nat.ptr(mean), nat.ptr(invstd), int(ctx.relu), nat.ptr(dgamma), nat.ptr(dbeta),
                                                      nat.ptr(ws), nat.current_stream(dev)), "magat_bn_train_backward_f32")
        return dx, dgamma, dbeta, None, None, None, None, None


def _batch_norm(m, t, relu=False):
    """torch.nn.BatchNorm2d.forward (+ ReLU) over a pixel-major tensor [P][M][C]: the statistics run over every (pixel, agent)
    row, i.e. over (N, H, W) of the NCHW tensor; running statistics / num_batches_tracked are updated as nn.BatchNorm does.
    Training mode with affine parameters: the HIP kernels (one streaming pass per reduction); otherwise torch's batch_norm."""
    P, M, C = t.shape
    factor = 0.0 if m.momentum is None else m.momentum
    if m.training and m.track_running_stats and m.num_batches_tracked is not None:
        m.num_batches_tracked.add_(1)
        factor = 1.0 / float(m.num_batches_tracked) if m.momentum is None else m.momentum
    use_batch = m.training or (m.running_mean is None and m.running_var is None)
    if (use_batch and t.is_cuda and m.affine and t.dtype == torch.float32 and m.weight.dtype == torch.float32 and
            nat.lib().magat_bn_train_workspace_floats(P * M, C) > 0):
        track = m.training and m.track_running_stats
        y = _BatchNormTrain.apply(t.reshape(P * M, C), m.weight, m.bias, m.running_mean if track else None,
                                  m.running_var if track else None, factor, m.eps, relu)
        return y.view(P, M, C)
    y = _batch_norm_torch(m, t, factor, use_batch)
    return torch.relu(y) if relu else y


def _batch_norm_torch(m, t, factor, use_batch):
    P, M, C = t.shape
    y = tnf.batch_norm(t.reshape(P * M, C), m.running_mean if (not m.training or m.track_running_stats) else None,
                       m.running_var if (not m.training or m.track_running_stats) else None, m.weight, m.bias, use_batch,
                       factor, m.eps)
    return y.view(P, M, C)


def _conv(m, t, h, w):
    """nn.Conv2d `m` over the pixel-major map t [h*w][M][Cin4] -> (map, h', w')"""
    kh, kw = m.kernel_size
    s, p = m.stride[0], m.padding[0]
    y = _ConvPixelMajor.apply(t, m.weight, h, w, s, p)
    ho, wo = (h + 2 * p - kh) // s + 1, (w + 2 * p - kw) // s + 1
    if m.bias is not None:
        y = y + m.bias.view(1, 1, -1)
    return y, ho, wo


def _basic_block(blk, t, h, w):
    out, ho, wo = _conv(blk.conv1, t, h, w)
    out = _batch_norm(blk.bn1, out, relu=True)
    out, _, _ = _conv(blk.conv2, out, ho, wo)
    out = _batch_norm(blk.bn2, out)
    if blk.downsample is None:
        res = t
    else:
        res, _, _ = _conv(blk.downsample[0], t, h, w)
        res = _batch_norm(blk.downsample[1], res)
    return torch.relu(out + res), ho, wo


def resnet_forward(body, x):
    """resnet.ResNet / ResNetSlim forward (resnet_pytorch.py:495-524) on the HIP convolution kernels, differentiable.
    x (M, 3, H, W) CUDA float32 -> (M, num_classes, H', W') like body(x)."""
    M, c, h, w = x.shape
    t = _pad4(x.float().permute(2, 3, 0, 1).reshape(h * w, M, c)).contiguous()         # [pixel][agent][4]
    t, h, w = _conv(body.conv1, t, h, w)
    t = _batch_norm(body.bn1, t, relu=True)
    for i in range(body.n_layers):
        t, h, w = _basic_block(getattr(body, "layer%d" % (i + 1))[0], t, h, w)
    k = body.avgpool.kernel_size
    k = k if isinstance(k, int) else k[0]
    h2, w2 = h // k, w // k
    C = t.shape[2]
    t = t.view(h, w, M, C)[:h2 * k, :w2 * k].reshape(h2, k, w2, k, M, C).mean(dim=(1, 3)).reshape(h2 * w2, M, C)
    t, _, _ = _conv(body.fc, t.contiguous(), h2, w2)
    return t.permute(1, 2, 0).reshape(M, t.shape[2], h2, w2).contiguous()      # (NCHW like body(x): callers .view() it)


def conv_stack_forward(seq, x):
    """CNN_mode 'Default' (decentralplanner_GAT_bottleneck.py:118-140: five conv3x3(+bias) + BatchNorm + ReLU layers with a
    MaxPool2d(2) behind the first, third and fifth) on the same kernels: any nn.Sequential of Conv2d / BatchNorm2d / ReLU /
    MaxPool2d / AvgPool2d / Dropout over pixel-major maps.  x (M, C, H, W) -> (M, C', H', W')."""
    import torch.nn as nn
    M, c, h, w = x.shape
    t = _pad4(x.float().permute(2, 3, 0, 1).reshape(h * w, M, c)).contiguous()
    mods = list(seq)
    fused = set()
    for i, m in enumerate(mods):
        if isinstance(m, nn.Conv2d):
            t, h, w = _conv(m, t, h, w)
        elif isinstance(m, nn.BatchNorm2d):
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)          # (BatchNorm + ReLU: one kernel pair)
            if relu:
                fused.add(i + 1)
            t = _batch_norm(m, t, relu=relu)
        elif isinstance(m, nn.ReLU):
            if i not in fused:
                t = torch.relu(t)
        elif isinstance(m, (nn.MaxPool2d, nn.AvgPool2d)):
            k = m.kernel_size if isinstance(m.kernel_size, int) else m.kernel_size[0]
            st = m.stride if isinstance(m.stride, int) else m.stride[0]
            if st != k or (m.padding if isinstance(m.padding, int) else m.padding[0]) != 0:
                raise NotImplementedError("pooling with stride != kernel or padding")
            h2, w2, C = h // k, w // k, t.shape[2]
            v = t.view(h, w, M, C)[:h2 * k, :w2 * k].reshape(h2, k, w2, k, M, C)
            t = (v.amax(dim=(1, 3)) if isinstance(m, nn.MaxPool2d) else v.mean(dim=(1, 3))).reshape(h2 * w2, M, C).contiguous()
            h, w = h2, w2
        elif isinstance(m, nn.Dropout):
            t = m(t)
        else:
            raise NotImplementedError("layer %r in a convolution stack" % (type(m).__name__,))
    return t.permute(1, 2, 0).reshape(M, t.shape[2], h, w).contiguous()


def _is_conv_stack(seq):
    import torch.nn as nn
    mods = list(seq)
    return len(mods) > 0 and isinstance(mods[0], nn.Conv2d) and all(
        isinstance(m, (nn.Conv2d, nn.BatchNorm2d, nn.ReLU, nn.MaxPool2d, nn.AvgPool2d, nn.Dropout)) for m in mods) and all(
        m.out_channels % 32 == 0 and m.groups == 1 and m.dilation == (1, 1) and m.stride[0] == m.stride[1] and
        m.stride[0] <= 2 and m.padding_mode == "zeros" and not isinstance(m.padding, str) and
        m.padding[0] == m.padding[1] for m in mods if isinstance(m, nn.Conv2d)) and all(
        not m.ceil_mode for m in mods if isinstance(m, (nn.MaxPool2d, nn.AvgPool2d)))


# Agents (rows of the convolution GEMMs) from which the HIP convolution / BatchNorm kernels beat torch's (MIOpen) in a training
# step.  Measured (bench.py `train_step`, profiles/r04j): 640 agents (the reference's own training batch, scripts/
# train_DMap.sh:30-46: 64 x 10) 3.83 ms on the HIP kernels against 3.14 ms on torch's - a step there is ~300 launches of a few
# microseconds each and torch's fused BatchNorm wins; 6400 agents 9.1 against 13.5 ms.  MAGAT_TRAIN_CNN = hip | torch forces
# one side; the default `auto` takes torch's convolutions below the crossover, so that dropping the module into the reference's
# training loop never makes a step slower.
TRAIN_HIP_MIN_AGENTS = 2048


def _use_hip_convs(x):
    import os
    mode = os.environ.get("MAGAT_TRAIN_CNN", "auto")
    if not x.is_cuda or mode == "torch":
        return False
    return mode == "hip" or x.shape[0] >= TRAIN_HIP_MIN_AGENTS


def convlayers_forward(conv_layers, x):
    """planner.ConvLayers(x) under autograd: the ResNet trunk on the HIP kernels when the input is on the GPU and the batch is
    large enough for them to win (see TRAIN_HIP_MIN_AGENTS; environment MAGAT_TRAIN_CNN = hip | torch | auto), the layers
    behind it (Dropout, Flatten, Linear) as they are."""
    from .resnet import _ResNetBase
    body = conv_layers[0] if len(conv_layers) > 0 else None
    if not _use_hip_convs(x):
        return conv_layers(x)
    if isinstance(body, _ResNetBase):
        y = resnet_forward(body, x)
        for m in list(conv_layers)[1:]:
            y = m(y)
        return y
    if _is_conv_stack(conv_layers):
        return conv_stack_forward(conv_layers, x)
    return conv_layers(x)
