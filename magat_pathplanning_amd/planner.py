"""DecentralPlannerGATNet -- drop-in for the class of the same name in the reference's
graphs/models/decentralplanner_GAT_bottleneck{,_SkipConcat,_SkipConcatGNN,_SkipAddGNN}.py and
decentralplanner_GAT.py (selected by config.bottleneckMode exactly like
agents/decentralplannerlocal_OnlineExpert_GAT.py:66-83 selects the file).

Same constructor (config object), addGSO(S), forward(x) -> (B*N, 5) logits and state_dict layout.
Inference (eval / no_grad) runs entirely on the gfx950 kernels behind include/magat_hip.h:
  ConvLayers + compressMLP  -> magat_encoder_forward_f32   (BN folded; fp32-class f16x3 split products on the 16-bit
                               matrix cores with fp32 accumulation, LDS-resident BasicBlock chains, range-guarded with a
                               stream-ordered fp32-MFMA re-run; the pooled head and compressMLP on fp32 MFMA)
  GFL                       -> magat_gat_forward_packed_f32 (per-agent maps on MFMA + LDS / wave-softmax graph kernel; CSR
                               kernels for N > 128 or bf16 storage)
  actionsMLP (+skip inputs) -> magat_conv_gemm_f32          (skip source as second K segment)
Training (autograd on): the graph layer's forward AND backward run on HIP kernels (graphml._GatTrainFunction,
_GnnTrainFunction), and so do the convolutions of the ResNet trunks (train_cnn.py: forward, input and weight gradients on the
float32 matrix-core kernels); BatchNorm / ReLU / pooling and the MLPs train through torch ops on the GPU.
"""
import ctypes
import hashlib
import operator
import os
import types

import torch
import torch.nn as nn

from . import _native as nat
from . import encoder as enc
from .graphml import (_MODES, CsrStructure, dense_route, GraphFilterBatchAttentional, GraphFilterBatchAttentional_Origin,
                      gat_forward_rows)
from .resnet import ResNet, ResNetSlim

_SKIP_FILES = {
    "BottomNeck_only": "only",
    "BottomNeck_skipConcat": "skipConcat",
    "BottomNeck_skipConcatGNN": "skipConcatGNN",
    "BottomNeck_skipAddGNN": "skipAddGNN",
}


def weights_init(m):
    """graphs/weights_initializer.py:11-23."""
    name = m.__class__.__name__
    if name.find("Conv") != -1:
        nn.init.xavier_normal_(m.weight)
    elif name.find("BatchNorm") != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0.0)
    elif name.find("Linear") != -1:
        nn.init.xavier_normal_(m.weight)
        m.bias.data.fill_(0.0)


CAL_AGENTS = 2048      # agents of the canonical calibration batch (synthetic.calibration_states)


from .train_cnn import convlayers_forward  # noqa: E402

_DATA_PTR = torch.Tensor.data_ptr
_VERSION = operator.attrgetter("_version")
# Registration epoch: bumped whenever ANY module of the process registers a parameter, buffer or submodule (torch's global
# registration hooks; `m.weight = nn.Parameter(..)` and `seq[0] = nn.Linear(..)` go through them) - what tells _weights_key that
# its cached tensor list may no longer be the module tree's.  Without the hooks (an older torch) the list is re-derived per call.
_REG_EPOCH = [0]


def _bump_reg_epoch(*_a):
    _REG_EPOCH[0] += 1


def _install_registration_hooks():
    from torch.nn.modules import module as tm
    names = ("register_module_parameter_registration_hook", "register_module_buffer_registration_hook",
             "register_module_module_registration_hook")
    if not all(hasattr(tm, n) for n in names):
        return False
    for n in names:
        getattr(tm, n)(_bump_reg_epoch)
    return True


_REG_HOOKED = _install_registration_hooks()


class _Runtime:
    """Device-side caches of one module instance (never pickled)."""

    def __init__(self):
        self.key = None
        self.pack = None
        self.desc = None
        self.act = None
        self.buffers = {}
        self.ws = None
        self.csr = CsrStructure()      # CSR + CSC structure of the GSO (large graphs / bf16 storage), made at addGSO
        self.plan = None               # step plan of the last plain forward (DecentralPlannerGATNet._plan_build)
        self.calibrated = True         # activation scales of the split arithmetic folded for the current weights
        self.act_scales = None
        self.digest = None             # fingerprint of the folded encoder pack (what a calibration belongs to)
        self.scaled_at = 0
        self.pack_host = self.pack_offs = self.pack_meta = None


class DecentralPlannerGATNet(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.S = None
        self.numAgents = config.num_agents
        self.skip = _SKIP_FILES.get(getattr(config, "bottleneckMode", ""), "legacy")
        inW = inH = config.FOV + 2
        numAction = 5
        bottleneck = config.bottleneckFeature if self.skip != "legacy" else config.numInputFeatures

        mode = config.CNN_mode
        if self.skip == "skipAddGNN" and mode in ("ResNetSlim_withMLP", "ResNetLarge_withMLP"):
            mode = "Default"    # that reference file has no *_withMLP branch (…SkipAddGNN.py:90-117)
        self.cnn_mode = mode
        if mode in ("ResNetSlim_withMLP", "ResNetLarge_withMLP"):
            body = ResNetSlim() if "Slim" in mode else ResNet()
            self.ConvLayers = nn.Sequential(body, nn.Dropout(0.2), nn.Flatten(),
                                            nn.Linear(1152, config.numInputFeatures, bias=True))
            numFeatureMap = config.numInputFeatures
        elif mode in ("ResNetSlim", "ResNetLarge"):
            body = ResNetSlim() if "Slim" in mode else ResNet()
            self.ConvLayers = nn.Sequential(body, nn.Dropout(0.2))
            numFeatureMap = 1152
        else:
            chans = [3, 32, 32, 64, 64, 128]
            layers, w, h = [], inW, inH
            for l in range(5):
                layers += [nn.Conv2d(chans[l], chans[l + 1], 3, 1, 1, bias=True), nn.BatchNorm2d(chans[l + 1]),
                           nn.ReLU(inplace=True)]
                if l % 2 == 0:
                    layers.append(nn.MaxPool2d(kernel_size=2))
                    w, h = (w - 2) // 2 + 1, (h - 2) // 2 + 1
            self.ConvLayers = nn.Sequential(*layers)
            numFeatureMap = chans[-1] * w * h
        self.numFeatureMap = numFeatureMap
        self.compressMLP = nn.Sequential(nn.Linear(numFeatureMap, bottleneck, bias=True), nn.ReLU(inplace=True))
        self.numFeatures2Share = bottleneck

        self.L = 1
        self.F = [bottleneck, bottleneck]
        self.K = [config.nGraphFilterTaps]
        self.P = [config.nAttentionHeads]
        self.E = 1
        self.bias = True
        if config.attentionMode not in ("GAT_modified", "KeyQuery", "GAT_origin"):
            raise NotImplementedError("attentionMode %r is outside the built hot path (SURVEY.md section 8(f))"
                                      % (config.attentionMode,))
        layer_cls = GraphFilterBatchAttentional_Origin if config.attentionMode == "GAT_origin" else GraphFilterBatchAttentional
        self.GFL = nn.Sequential(layer_cls(
            self.F[0], self.F[1], self.K[0], self.P[0], self.E, self.bias,
            concatenate=config.AttentionConcat, attentionMode=config.attentionMode))
        # optional key (not in the reference's configs): HBM storage type inside the GAT layer at inference,
        # 'fp32' (default, the 1e-4 parity path) or 'bf16' (BASELINE config 5)
        storage = getattr(config, "gat_storage", "fp32")
        if storage not in ("fp32", "bf16"):
            raise ValueError("config.gat_storage must be 'fp32' or 'bf16', got %r" % (storage,))
        if storage == "bf16":
            self.GFL[0].storage_dtype = torch.bfloat16

        width = self.F[-1] * config.nAttentionHeads if config.AttentionConcat else self.F[-1]
        self.gat_width = width
        if self.skip == "skipAddGNN" and width != bottleneck:
            # the reference's torch.add of (B*N, G) and (B*N, P*F) raises at the first forward
            # (decentralplanner_GAT_bottleneck_SkipAddGNN.py:303-307); say so at construction instead of reading the
            # actionsMLP weights with the wrong K split
            raise RuntimeError("BottomNeck_skipAddGNN adds the %d-wide bottleneck feature to the %d-wide graph-layer output: "
                               "needs AttentionConcat=False or nAttentionHeads=1" % (bottleneck, width))
        if self.skip == "skipConcat":
            width += numFeatureMap
        elif self.skip == "skipConcatGNN":
            width += bottleneck
        if config.use_dropout:
            self.actionsMLP = nn.Sequential(nn.Linear(width, config.numInputFeatures), nn.ReLU(inplace=True),
                                            nn.Dropout(p=0.2), nn.Linear(config.numInputFeatures, numAction),
                                            nn.Dropout(p=0.2))
        else:
            self.actionsMLP = nn.Sequential(nn.Linear(width, numAction))
        self.apply(weights_init)
        self._rt = _Runtime()
        self._flat, self._flat_age, self._flat_epoch = None, 0, -1     # cached tensor list of _weights_key
        # agent count the batch-size-dependent kernel forms are chosen on (0: each call's own); set around a shard's forward by
        # distributed.sharded_forward (magat_encoder_desc.form_agents)
        self.form_agents = 0
        # host side of the step: reuse everything that does not change between two forwards of the same shape (_plan_build);
        # False = resolve buffers / workspaces / arguments on every call (the general path the plan is tested against)
        self.step_plan = True
        # layer magnitudes the activation scales are folded from: {"digest", "absmax", "source"}.  Unlike _rt it IS pickled: a
        # spawned worker that unpickles the same weights folds the same exponents without measuring anything
        self._cal = None

    # ------------------------------------------------------------------ pickling (spawned workers)
    def __getstate__(self):
        st = self.__dict__.copy()
        st["_rt"] = None
        st["S"] = None
        st["_flat"] = None
        return st

    def __setstate__(self, st):
        super().__setstate__(st)
        self._rt = _Runtime()
        self._flat, self._flat_age, self._flat_epoch = None, 0, -1
        self.__dict__.setdefault("_cal", None)       # (modules pickled before the calibration record existed)
        self.__dict__.setdefault("step_plan", True)

    # ------------------------------------------------------------------ boundary
    def addGSO(self, S):
        """…bottleneck.py:262-278: aliases the caller's tensor and scrubs it in place."""
        assert len(S.shape) == 3
        scrub = self.skip in ("only", "legacy")        # only these reference files zero NaNs
        gso_mode = {"dist_GSO_one": 1, "full_GSO": 2}.get(self.config.GSO_mode, 0)
        if S.is_cuda and S.is_contiguous() and S.dtype in (torch.float32, torch.float64) and gso_mode != 2:
            layer = self.GFL[0]
            B_, N_ = S.shape[0], S.shape[1]
            if (S.numel() > 0 and S.shape[1] == S.shape[2] and not self.training and
                    (layer.storage_dtype == torch.bfloat16 or
                     not dense_route(N_, layer)) and CsrStructure.supported(B_, N_)):
                # large graph / bf16 storage: the layer runs on the CSR kernels.  ONE pass over S does the scrub and leaves
                # the bit matrix the CSR + CSC structure is built from (no host synchronisation, nothing re-read later)
                self._rt.csr.build(S, 1 if layer.attentionMode == "GAT_origin" else 0, scrub_nan=scrub, gso_mode=gso_mode)
                self.S = S.unsqueeze(1)
                return
            self._rt.csr.key = None
            if (scrub or gso_mode) and S.numel() > 0:        # an empty GSO is accepted here, like the reference's addGSO
                with torch.cuda.device(S.device):
                    nat.check(nat.lib().magat_gso_prepare(nat.ptr(S), 1 if S.dtype == torch.float64 else 0,
                                                          S.numel(), 1 if scrub else 0, gso_mode,
                                                          nat.current_stream(S.device)), "magat_gso_prepare")
            self.S = S.unsqueeze(1)
            return
        self.S = S.unsqueeze(1)
        if scrub:
            self.S[torch.isnan(self.S)] = 0
        if gso_mode == 1:
            self.S[self.S > 0] = 1
        elif gso_mode == 2:
            self.S = torch.ones_like(self.S).to(self.config.device)

    def returnAttentionGSO(self):
        return self.GFL[0].returnAttentionGSO()

    def range_status(self):
        """Range guard of the split arithmetic (include/magat_hip.h): did the LAST inference forward leave the range the
        f16 planes carry exactly, so that the encoder / the graph layer's maps were re-run on the float32 MFMA kernels
        (same stream, automatic), and how often has that happened since the workspaces were allocated.  Synchronises."""
        out = {"encoder_rerun": False, "encoder_reruns": 0, "gat_rerun": False, "gat_reruns": 0,
               "act_scales": None if self._rt is None else self._rt.act_scales}
        lib = nat.lib()
        st = (ctypes.c_int32 * 2)()
        rt = self._rt
        if rt is not None and rt.ws is not None:
            with torch.cuda.device(rt.ws.device):
                nat.check(lib.magat_encoder_read_status(nat.ptr(rt.ws), st, nat.current_stream(rt.ws.device)),
                          "magat_encoder_read_status")
            out["encoder_rerun"], out["encoder_reruns"] = bool(st[0]), int(st[1])
        ws = self.GFL[0]._scratch.workspace
        if ws is not None and self.GFL[0].storage_dtype != torch.bfloat16:
            with torch.cuda.device(ws.device):
                nat.check(lib.magat_gat_read_status(nat.ptr(ws), st, nat.current_stream(ws.device)),
                          "magat_gat_read_status")
            out["gat_rerun"], out["gat_reruns"] = bool(st[0]), int(st[1])
        sc = out["act_scales"]
        if out["encoder_reruns"] >= 3 and sc and sc.get("source") == "canonical":
            # the canonical calibration batch does not fit this deployment's inputs (non-binary channels, another FOV encoding):
            # every forward then pays the float32 re-run - correct, but slow.  Say so once; calibrate(x) on real inputs fixes it
            out["hint"] = ("the float32 re-run fired %d times with activation scales from the canonical calibration batch: "
                           "call model.calibrate(x) on representative inputs (and again after load_state_dict / training)"
                           % out["encoder_reruns"])
            if not getattr(self, "_warned_reruns", False):
                import warnings
                warnings.warn("magat_pathplanning_amd: " + out["hint"])
                self._warned_reruns = True
        return out

    def forward(self, inputTensor):
        (B, N, C, W, H) = inputTensor.shape
        dev = torch.device(self.config.device)
        x = inputTensor.reshape(B * N, C, W, H).to(dev)
        if self.S is None:
            raise TypeError("addGSO must be called before forward")
        side = self.config.FOV + 2
        if (C, W, H) != (3, side, side):
            # the reference fails in its first Linear (mat1 and mat2 shapes cannot be multiplied); the folded encoder is
            # built for one map size, so say it up front instead of reading the tensor with the wrong geometry
            raise RuntimeError("DecentralPlannerGATNet built for (3, %d, %d) state maps (config.FOV + 2), got (%d, %d, %d)"
                               % (side, side, C, W, H))
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if needs_grad or self.training:
            nat.require_device_or_composite(x, "DecentralPlannerGATNet in training / autograd mode")
            return self._forward_autograd(x, B, N)
        return self._forward_hip(x, B, N)

    # ------------------------------------------------------------------ training path (torch ops)
    def _forward_autograd(self, x, B, N):
        feat = convlayers_forward(self.ConvLayers, x)       # (ResNet trunks: HIP convolution kernels, train_cnn.py)
        feat = feat.view(feat.size(0), -1)
        comp = self.compressMLP(feat)
        xg = comp.reshape(B, N, self.numFeatures2Share).permute(0, 2, 1)
        self.GFL[0].addGSO(self.S)
        shared = self.GFL(xg)
        shared = shared.permute(0, 2, 1).reshape(B * N, shared.shape[1])
        if self.skip == "skipConcat":
            shared = torch.cat((feat, shared), dim=1)
        elif self.skip == "skipConcatGNN":
            shared = torch.cat((comp, shared), dim=1)
        elif self.skip == "skipAddGNN":
            shared = torch.add(comp, shared)
        return self.actionsMLP(shared)

    # ------------------------------------------------------------------ inference path (HIP)
    def _walk_tensors(self):
        """Every parameter and buffer of the module tree, in a fixed walk order (an iterative walk over the module dicts:
        nn.Module.parameters() / buffers() are recursive generators with a memo set, several times slower)."""
        flat = []
        stack = [self]
        while stack:
            m = stack.pop()
            for t in m._parameters.values():
                if t is not None:
                    flat.append(t)
            for t in m._buffers.values():
                if t is not None:
                    flat.append(t)
            for c in m._modules.values():
                if c is not None:
                    stack.append(c)
        return flat

    def _weights_key(self, dev):
        """(device, addresses, versions) of every parameter and buffer of the module tree: what the folded / packed weights of
        the HIP path were made from.  It sits on the critical path of the closed-loop step (nothing can be launched before
        it), so the tensors are visited through a cached flat list with two C-level maps (6 us instead of 18 for the ~70
        tensors of the reference's model).  Changes under any of: load_state_dict / optimizer steps / in-place updates
        (version), .to() / .data assignment (address), a replaced Parameter / buffer / submodule anywhere in the tree (the
        registration epoch above: the list is re-derived).  The list is also re-derived on _apply() / load_state_dict()
        (overridden below), after unpickling and on every 64th call (the backstop for surgery on the module dicts themselves:
        `del m._parameters[..]`; `invalidate_weights()` makes the next forward see that at once)."""
        flat = self._flat
        self._flat_age += 1
        if flat is None or self._flat_epoch != _REG_EPOCH[0] or (self._flat_age & 63) == 0 or not _REG_HOOKED:
            walked = self._walk_tensors()
            self._flat_epoch = _REG_EPOCH[0]
            if flat is None or len(walked) != len(flat) or any(a is not b for a, b in zip(walked, flat)):
                flat = self._flat = walked
        return (dev, tuple(map(_DATA_PTR, flat)), tuple(map(_VERSION, flat)))

    def invalidate_weights(self):
        """Forget the cached tensor list and the folded weights: the next forward re-derives everything from the module tree."""
        self._flat = None
        if self._rt is not None:
            self._rt.key = None

    def _apply(self, fn, *a, **kw):
        self._flat = None
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self._flat = None
        return super().load_state_dict(*a, **kw)

    def _refresh(self, dev):
        rt = self._rt
        key = self._weights_key(dev)
        if rt.key == key:
            return rt
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        rt.pack = rt.desc = None
        cells = None                   # CNN_mode Default: pooled cells of the last map (feature order, see below)
        if self.cnn_mode.startswith("ResNet"):
            lin = (sd["ConvLayers.3.weight"], sd["ConvLayers.3.bias"]) if self.cnn_mode.endswith("_withMLP") else None
            pack, offs, meta = enc.fold_resnet(sd, self.config.FOV + 2, self.config.FOV + 2, "ConvLayers.0", lin,
                                               (sd["compressMLP.0.weight"], sd["compressMLP.0.bias"]))
            # room for the activation-scale block behind the pack (filled by the calibration pass of the first forward)
            rt.scaled_at = pack.numel()
            rt.pack_host, rt.pack_offs, rt.pack_meta = pack, offs, meta
            rt.pack = torch.cat((pack, torch.zeros(enc.SCALED_BLOCK_FLOATS))).to(dev)
            rt.calibrated, rt.act_scales = False, None
            rt.digest = hashlib.blake2b(pack.numpy().tobytes(), digest_size=16).hexdigest()
            self.GFL[0]._scratch.x_scale = 0.0
            d = nat.EncoderDesc()
            d.variant, d.H, d.W = meta["variant"], meta["H"], meta["W"]
            d.n_feat, d.n_comp = meta["n_feat"], meta["n_comp"]
            d.pack = rt.pack.data_ptr()
            for i, o in enumerate(offs):
                d.off[i] = o
            d.chain_off = meta.get("chain", 0)
            d.chain3_off = meta.get("chain3", 0)
            d.head16_off = meta.get("head16", 0)
            d.comp16_off = meta.get("comp16", 0)
            d.l1frag_off = meta.get("l1frag", 0)
            d.headfrag_off = meta.get("headfrag", 0)
            d.compfrag_off = meta.get("compfrag", 0)
            d.scaled_off = 0
            rt.desc = d
        else:
            side = self.config.FOV + 2
            if getattr(self, "dilated_version", 0):
                pack, offs, meta = enc.fold_dilated_cnn(sd, self.dilated_version, side, side, "ConvLayers",
                                                        (sd["compressMLP.0.weight"], sd["compressMLP.0.bias"]))
            else:
                pack, offs, meta = enc.fold_default_cnn(sd, side, side, "ConvLayers",
                                                        (sd["compressMLP.0.weight"], sd["compressMLP.0.bias"]))
            rt.pack = pack.to(dev)
            d = nat.EncoderDesc()
            d.variant, d.H, d.W = meta["variant"], meta["H"], meta["W"]
            d.n_feat, d.n_comp = meta["n_feat"], meta["n_comp"]
            d.pack = rt.pack.data_ptr()
            for i, o in enumerate(offs):
                d.off[i] = o
            rt.desc = d
            cells = meta["cells"]
        # actionsMLP first layer: [w_skip | w_gat] -> in (skip source) + in2 (GAT output) K segments
        w0 = sd["actionsMLP.0.weight"].to(dev, torch.float32)
        if cells is not None and cells != (1, 1) and self.skip == "skipConcat":
            # Default CNN at a map size with several pooled cells: the HIP encoder stores the feature map (cell, channel)-
            # ordered, the skip-concat half of actionsMLP reads it through permuted columns (encoder.fold_default_cnn)
            nfm = self.numFeatureMap
            w0 = torch.cat((enc.cell_major_columns(w0[:, :nfm], *cells), w0[:, nfm:]), dim=1)
        if self.skip == "skipAddGNN":
            w0 = torch.cat((w0, w0), dim=1)
        rt.act = [w0.contiguous(), sd["actionsMLP.0.bias"].to(dev, torch.float32).contiguous()]
        if self.config.use_dropout:
            rt.act += [sd["actionsMLP.3.weight"].to(dev, torch.float32).contiguous(),
                       sd["actionsMLP.3.bias"].to(dev, torch.float32).contiguous()]
        rt.cw = sd["compressMLP.0.weight"].to(dev, torch.float32).contiguous()
        rt.cb = sd["compressMLP.0.bias"].to(dev, torch.float32).contiguous()
        rt.key = key
        return rt

    def _measure(self, rt, x, dev):
        """magat_encoder_calibrate_f32 over the agent rows x (M, 3, W, H): ONE float32 pass (the guard's layer-by-layer
        kernels) that leaves the largest |output| of every layer in absmax[16].  Own scratch: nothing of the forward's
        buffers is touched.  Returns the DEVICE tensor (no synchronisation here)."""
        lib = nat.lib()
        M = x.shape[0]
        with torch.cuda.device(dev):
            stream = nat.current_stream(dev)
            ws = torch.empty(lib.magat_encoder_workspace_bytes(ctypes.byref(rt.desc), M), dtype=torch.uint8, device=dev)
            ws[:256].zero_()
            feat = torch.empty(M, self.numFeatureMap, dtype=torch.float32, device=dev)
            comp = torch.empty(M, self.numFeatures2Share, dtype=torch.float32, device=dev)
            absmax = torch.zeros(16, dtype=torch.float32, device=dev)
            nat.check(lib.magat_encoder_calibrate_f32(ctypes.byref(rt.desc), nat.ptr(x), nat.ptr(feat), self.numFeatureMap,
                                                      nat.ptr(comp), self.numFeatures2Share, nat.ptr(ws), ws.numel(), M,
                                                      nat.ptr(absmax), stream), "magat_encoder_calibrate_f32")
        return absmax

    def _fold_scales(self, rt, dev):
        """self._cal["absmax"] -> the scale block (encoder.fold_activation_scales) behind the pack and the graph layer's
        input scale: host arithmetic on 16 floats, the same in every process that holds the same weights and magnitudes."""
        a = list(self._cal["absmax"])
        blk, info = enc.fold_activation_scales(rt.pack_host, rt.pack_offs, rt.pack_meta, a) if rt.pack_meta.get("chain3", 0) \
            else (None, {})
        if blk is not None:
            rt.pack[rt.scaled_at:rt.scaled_at + enc.SCALED_BLOCK_FLOATS].copy_(blk.to(dev))
            rt.desc.scaled_off = rt.scaled_at
        # the graph layer's input (comp): scaled UP only (target 2^6: Q = W x and U = H x keep headroom in their own planes).
        # A layer input beyond the planes' range is left to the range guard: its scores x_i W x_j are then ~1e10 with float32
        # steps of ~1e3 - only the float32 form, which rounds in the reference's own order, still follows the reference there
        e_x = max(0, enc.scale_exponent(a[8], target_log2=6))
        self.GFL[0]._scratch.x_scale = 2.0 ** e_x
        info.update(gat_in=e_x, absmax=[float(v) for v in a[:9]], source=self._cal["source"])
        rt.act_scales = info
        rt.calibrated = True

    def _ensure_calibrated(self, rt, dev):
        """Activation scales for the current weights (include/magat_hip.h "Activation scales").  The magnitudes come from
        (1) self._cal when it belongs to these weights (an explicit calibrate(), or inherited through pickling), else
        (2) ONE float32 pass over the canonical batch synthetic.calibration_states(FOV, CAL_AGENTS, seed 0): a function of the
        config alone, so every rank / spawned worker / fresh module with the same state_dict folds the SAME exponents -
        the logits of a planning instance do not depend on which batch or shard a process happened to see first (SURVEY.md
        section 8(e): shards concatenate bit-exactly to the single-process result).  One host synchronisation per set of
        weights.  Inputs that drive a layer 64x beyond the measured magnitude leave the planes and are re-run in float32 by
        the range guard (range_status() counts it); calibrate(x) measures on the caller's own data instead."""
        if self._cal is None or self._cal.get("digest") != rt.digest:
            from .synthetic import calibration_states
            xc = calibration_states(self.config.FOV, CAL_AGENTS, seed=0).to(dev)
            self._cal = {"digest": rt.digest, "absmax": self._measure(rt, xc, dev).cpu().tolist(), "source": "canonical"}
        self._fold_scales(rt, dev)

    @torch.no_grad()
    def calibrate(self, x=None, group=None):
        """Explicit calibration of the activation scales (optional; the default is the canonical batch, see
        _ensure_calibrated).  x: state tensors (B, N, 3, W, H) or (M, 3, W, H) on the model's device, None = the canonical
        batch.  With torch.distributed initialised the measured magnitudes are all-reduced (MAX) over `group`, so that ranks
        that calibrate on their own shards still agree on every exponent.  The result stays with the module (it is pickled
        with it) until the weights change; returns range_status()["act_scales"]."""
        import torch.distributed as dist
        dev = torch.device(self.config.device)
        rt = self._refresh(dev)
        if rt.desc.variant not in (0, 1) or os.environ.get("MAGAT_ACT_SCALE", "1") != "1":
            return None
        if x is None:
            from .synthetic import calibration_states
            xr, source = calibration_states(self.config.FOV, CAL_AGENTS, seed=0).to(dev), "canonical"
        else:
            side = self.config.FOV + 2
            xr, source = x.reshape(-1, 3, side, side).to(dev).contiguous().float(), "user"
        absmax = self._measure(rt, xr, dev)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            red = absmax if dist.get_backend(group) != "gloo" else absmax.cpu()
            dist.all_reduce(red, op=dist.ReduceOp.MAX, group=group)
            absmax = red
        self._cal = {"digest": rt.digest, "absmax": absmax.cpu().tolist(), "source": source}
        self._fold_scales(rt, dev)
        return rt.act_scales

    def _run_encoder(self, rt, x, M, dev, stream):
        """ConvLayers + compressMLP of M agents on the HIP encoder: (feat (M, numFeatureMap), comp (M, G))."""
        lib = nat.lib()
        G, nfm = self.numFeatures2Share, self.numFeatureMap
        feat = self._buf("feat", (M, nfm), dev)
        comp = self._buf("comp", (M, G), dev)
        need = lib.magat_encoder_workspace_bytes(ctypes.byref(rt.desc), M)
        if rt.ws is None or rt.ws.numel() < need or rt.ws.device != dev:
            rt.ws = torch.empty(need, dtype=torch.uint8, device=dev)
            rt.ws[:256].zero_()          # range-guard status block (magat_encoder_read_status)
        if not rt.calibrated and rt.desc.variant in (0, 1) and os.environ.get("MAGAT_ACT_SCALE", "1") == "1":
            # first forward with these weights: the power-of-two activation scales of the split arithmetic are folded
            # from the canonical calibration batch (or from an explicit / inherited calibration) - never from the batch
            # at hand, so the forward runs like every later one and like every other process with these weights
            self._ensure_calibrated(rt, dev)
        # a shard of a larger batch chooses its batch-size-dependent kernel forms on the GLOBAL agent count (set by
        # distributed.sharded_forward): the shards then concatenate to the single-process result bit for bit
        rt.desc.form_agents = max(0, int(getattr(self, "form_agents", 0) or 0))
        # bf16 storage inside the graph layer: compressMLP's rows leave the encoder as bf16 too (ABI 7: from the epilogue that
        # produces them - no cast pass between the two layers)
        comp16 = None
        if getattr(self.GFL[0], "storage_dtype", None) == torch.bfloat16 and G % 4 == 0 and rt.desc.n_comp == G:
            comp16 = self._buf16("comp16", (M, G), dev)
        rt.desc.comp_bf16 = comp16.data_ptr() if comp16 is not None else None
        rt.comp16 = comp16
        nat.check(lib.magat_encoder_forward_f32(ctypes.byref(rt.desc), nat.ptr(x), nat.ptr(feat), nfm,
                                                nat.ptr(comp), G, nat.ptr(rt.ws), rt.ws.numel(), M, stream),
                  "magat_encoder_forward_f32")
        rt.desc.comp_bf16 = None          # (the pointer belongs to this call's buffers)
        return feat, comp

    def _run_actions(self, rt, feat, comp, gat, gat_rows, M, dev, stream):
        """actionsMLP (+ the skip source as a second K segment) on the graph layer's rows."""
        lib = nat.lib()
        nout = rt.act[0].shape[0]
        out = torch.empty(M, nout, dtype=torch.float32, device=dev)
        d = nat.ConvGemmDesc()
        rows16 = gat_rows.dtype == torch.bfloat16
        if self.skip in ("skipConcat", "skipConcatGNN", "skipAddGNN"):
            src = feat if self.skip == "skipConcat" else comp
            d.inp, d.Cin, d.lda = src.data_ptr(), src.shape[1], src.stride(0)
            d.in2, d.C2, d.lda2 = gat_rows.data_ptr(), gat_rows.shape[1], gat_rows.stride(0)
            d.W2, d.stride2 = 1, 1
            d.bf16_rows = 2 if rows16 else 0
        else:
            d.inp, d.Cin, d.lda = gat_rows.data_ptr(), gat_rows.shape[1], gat_rows.stride(0)
            d.bf16_rows = 1 if rows16 else 0
        d.wt, d.bias, d.out = rt.act[0].data_ptr(), rt.act[1].data_ptr(), out.data_ptr()
        d.M, d.Hin, d.Win, d.kH, d.kW, d.stride, d.pad, d.Hout, d.Wout = M, 1, 1, 1, 1, 1, 0, 1, 1
        d.Cout, d.ldc, d.relu = nout, nout, 1 if self.config.use_dropout else 0
        d.tag = nat.TAG_ACTIONS
        rc = lib.magat_conv_gemm_f32(ctypes.byref(d), stream)
        if rc == -2 and rows16:
            # MAGAT_ERR_UNSUPPORTED: only the streamed-dot-product kernel reads bf16 rows, and it declined (its weight
            # block exceeds 64 KB of LDS: e.g. CNN_mode Default with skipConcat at a large FOV): widen the rows and take
            # the float32 kernel instead of failing the forward
            nat.check(lib.magat_cast_rows(nat.ptr(gat_rows), nat.ptr(gat), 0, M, self.gat_width, self.gat_width,
                                          self.gat_width, stream), "magat_cast_rows")
            if d.C2:
                d.in2, d.lda2 = gat.data_ptr(), gat.stride(0)
            else:
                d.inp, d.lda = gat.data_ptr(), gat.stride(0)
            d.bf16_rows = 0
            rc = lib.magat_conv_gemm_f32(ctypes.byref(d), stream)
        nat.check(rc, "magat_conv_gemm_f32(actionsMLP.0)")
        if self.config.use_dropout:
            out2 = torch.empty(M, rt.act[2].shape[0], dtype=torch.float32, device=dev)
            nat.check(lib.magat_linear_f32(nat.ptr(out), nout, nat.ptr(rt.act[2]), nat.ptr(rt.act[3]),
                                           nat.ptr(out2), out2.shape[1], M, out2.shape[1], nout, 0, stream),
                      "magat_linear_f32(actionsMLP.3)")
            out = out2
        return out

    def _buf(self, name, shape, dev):
        t = self._rt.buffers.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.device != dev:
            t = torch.empty(shape, dtype=torch.float32, device=dev)
            self._rt.buffers[name] = t
        return t

    def _buf16(self, name, shape, dev):
        t = self._rt.buffers.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.device != dev:
            t = torch.empty(shape, dtype=torch.bfloat16, device=dev)
            self._rt.buffers[name] = t
        return t

    # ------------------------------------------------------------------ step plan (host side of the closed-loop step)
    def _plan_build(self, rt, B, N, dev):
        """Everything of a forward that does not change from step to step, resolved ONCE for (these weights, this batch shape,
        this device): buffers, workspaces, packed graph-layer weights, the ctypes arguments of the three C-ABI calls.  The
        closed-loop step of one planning instance is HOST-bound (device 58 us, Python + launches 100 us at N = 10,
        profiles/r06d): the general path re-derives all of this per call.  Built after a forward through the general path
        (which allocates and calibrates); None when the configuration is not the plain dense float32 one."""
        lib = nat.lib()
        layer = self.GFL[0]
        sc = layer._scratch
        G, nfm, M = self.numFeatures2Share, self.numFeatureMap, B * N
        if (layer.storage_dtype == torch.bfloat16 or self.config.use_dropout or rt.ws is None or sc.workspace is None or
                sc.packed is None or not rt.calibrated or not dense_route(N, layer) or
                self.skip not in ("skipConcat", "skipConcatGNN", "skipAddGNN", "only", "legacy")):
            return None
        b = rt.buffers
        feat, comp, gat = b.get("feat"), b.get("comp"), b.get("gat")
        if feat is None or comp is None or gat is None or feat.shape[0] != M or gat.shape[0] != M:
            return None
        pl = types.SimpleNamespace()
        pl.rt_key, pl.B, pl.N, pl.dev, pl.dev_index = rt.key, B, N, dev, dev.index
        pl.ws, pl.gws, pl.packed, pl.x_scale = rt.ws, sc.workspace, sc.packed, sc.x_scale
        pl.feat, pl.comp, pl.gat = feat, comp, gat
        pl.bias = None if layer.bias is None else layer.bias.detach().to(dev, torch.float32).reshape(-1).contiguous()
        F, K, P = layer.F, layer.K, layer.P
        mode = _MODES[layer.attentionMode]
        concat = 1 if layer.concatenate else 0
        pl.enc_tail = (nat.ptr(feat), nfm, nat.ptr(comp), G, nat.ptr(rt.ws), rt.ws.numel(), M)
        pl.gat_head = nat.ptr(comp)
        pl.gat_tail = (nat.ptr(sc.packed), nat.ptr(pl.bias), nat.ptr(gat), gat.stride(0), nat.ptr(sc.workspace),
                       sc.workspace.numel(), B, N, G, F, K, P, mode, concat)
        pl.tail_done = ctypes.c_int(0)
        pl.tail_done_ref = ctypes.byref(pl.tail_done)
        nout = rt.act[0].shape[0]
        d = nat.ConvGemmDesc()
        if self.skip in ("skipConcat", "skipConcatGNN", "skipAddGNN"):
            src = feat if self.skip == "skipConcat" else comp
            d.inp, d.Cin, d.lda = src.data_ptr(), src.shape[1], src.stride(0)
            d.in2, d.C2, d.lda2 = gat.data_ptr(), gat.shape[1], gat.stride(0)
            d.W2, d.stride2 = 1, 1
        else:
            d.inp, d.Cin, d.lda = gat.data_ptr(), gat.shape[1], gat.stride(0)
        d.wt, d.bias = rt.act[0].data_ptr(), rt.act[1].data_ptr()
        d.M, d.Hin, d.Win, d.kH, d.kW, d.stride, d.pad, d.Hout, d.Wout = M, 1, 1, 1, 1, 1, 0, 1, 1
        d.Cout, d.ldc, d.relu = nout, nout, 0
        d.tag = nat.TAG_ACTIONS
        pl.act, pl.act_ref, pl.nout = d, ctypes.byref(d), nout
        pl.desc_ref = ctypes.byref(rt.desc)
        return pl

    def _plan_step(self, pl, rt, x, M, dev):
        """One forward through a step plan: three C-ABI calls (encoder, graph layer, action head - two when the action head rode
        in the graph layer's last launch, magat_gat_forward_tail_f32) and one allocation (the logits).  Same kernels, same
        arguments as the general path."""
        lib = nat.lib()
        layer = self.GFL[0]
        S = self.S
        rt.desc.form_agents = max(0, int(self.form_agents or 0))
        rt.desc.comp_bf16 = None
        switch = torch.cuda.current_device() != pl.dev_index
        if switch:
            prev = torch.cuda.current_device()
            torch.cuda.set_device(pl.dev_index)
        try:
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            rc = lib.magat_encoder_forward_f32(pl.desc_ref, ctypes.c_void_p(x.data_ptr()), *pl.enc_tail, stream)
            if rc:
                nat.check(rc, "magat_encoder_forward_f32")
            layer.addGSO(S)
            # (the logits' buffer exists before the graph layer's call: the action head may ride in the layer's last launch)
            out = torch.empty(M, pl.nout, dtype=torch.float32, device=dev)
            pl.act.out = out.data_ptr()
            rc = lib.magat_gat_forward_tail_f32(pl.gat_head, ctypes.c_void_p(S.data_ptr()), 1 if S.dtype == torch.float64 else 0,
                                                *pl.gat_tail, pl.act_ref, pl.tail_done_ref, stream)
            if rc == -2:        # MAGAT_ERR_UNSUPPORTED: a library option that decides the layer's route (GAT_MFMA, GAT_WIDE_FROM)
                return None     # changed since the plan was built - the caller drops the plan and takes the general path
            if rc:
                nat.check(rc, "magat_gat_forward_tail_f32")
            layer.aij = None
            if not pl.tail_done.value:
                rc = lib.magat_conv_gemm_f32(pl.act_ref, stream)
                if rc:
                    nat.check(rc, "magat_conv_gemm_f32(actionsMLP.0)")
        finally:
            if switch:
                torch.cuda.set_device(prev)
        return out

    @torch.no_grad()
    def _forward_hip(self, x, B, N):
        if not x.is_cuda:
            raise nat.MagatNativeError("inference runs on the HIP path only; config.device=%r is not a GPU "
                                       "(no CPU fallback)" % (self.config.device,))
        lib = nat.lib()
        dev = x.device
        M = B * N
        rt = self._refresh(dev)
        x = x.contiguous().float()
        G = self.numFeatures2Share
        nfm = self.numFeatureMap
        layer0 = self.GFL[0]
        S = self.S
        plain = (S.is_cuda and S.shape[-1] == N and S.shape[0] == B and S.is_contiguous() and S.device == dev and
                 S.dtype in (torch.float32, torch.float64) and not layer0.return_attention and
                 not getattr(self.config, "return_attentionGSO", False))
        plain = plain and self.step_plan
        pl = rt.plan
        if (plain and pl is not None and pl.rt_key is rt.key and pl.B == B and pl.N == N and pl.dev == dev and pl.ws is rt.ws and
                pl.gws is layer0._scratch.workspace and pl.packed is layer0._scratch.packed and
                pl.x_scale == layer0._scratch.x_scale and rt.calibrated):
            out = self._plan_step(pl, rt, x, M, dev)
            if out is not None:
                return out
        rt.plan = None
        out = self._forward_general(rt, x, B, N, M, dev)
        if plain:
            rt.plan = self._plan_build(rt, B, N, dev)
        return out

    def _forward_general(self, rt, x, B, N, M, dev):
        lib = nat.lib()
        G = self.numFeatures2Share
        nfm = self.numFeatureMap
        with torch.cuda.device(dev):
            stream = nat.current_stream(dev)
            feat, comp = self._run_encoder(rt, x, M, dev, stream)
            layer = self.GFL[0]
            layer.addGSO(self.S)
            gat = self._buf("gat", (M, self.gat_width), dev)
            gat_rows = gat          # what the action head reads: `gat`, or the layer's bf16 rows (bf16 storage, see below)
            want_att = layer.return_attention or bool(getattr(self.config, "return_attentionGSO", False))
            Ns = self.S.shape[-1]
            if Ns > N:
                # more GSO nodes than agents: the graph layer zero-pads the signal to the GSO's size and trims its
                # output (graphML.py:4641-4646, 4670-4671) - rare (never in the published scripts), done with copies
                rows = torch.zeros(B, Ns, G, dtype=torch.float32, device=dev)
                rows[:, :N] = comp.view(B, N, G)
                full, aij = gat_forward_rows(rows.to(layer.storage_dtype) if layer.storage_dtype == torch.bfloat16 else rows,
                                             self.S, layer, want_attention=want_att)
                gat.copy_(full.view(B, Ns, -1)[:, :N, :self.gat_width].reshape(M, self.gat_width))
            elif layer.storage_dtype == torch.bfloat16:
                # bf16 storage inside the GAT layer (config.gat_storage='bf16', BASELINE config 5): the layer reads
                # and writes bf16 rows; the CNN/MLP GEMMs around it stay fp32
                comp16 = getattr(rt, "comp16", None)
                if comp16 is None or comp16.shape[0] != M:      # (an encoder that does not write the bf16 rows itself)
                    comp16 = self._buf16("comp16", (M, G), dev)
                    nat.check(lib.magat_cast_rows(nat.ptr(comp), nat.ptr(comp16), 1, M, G, G, G, stream), "magat_cast_rows")
                # the layer's bf16 rows go to the action head as they are when it runs as streamed dot products (at most 8
                # outputs, option SKINNY: its loader widens them); otherwise the layer's last kernel stores them widened
                # (device-built CSR + CSC structure), or a cast pass follows
                if rt.act[0].shape[0] <= 8 and nat.get_option("SKINNY") and self.gat_width % 8 == 0:
                    gat_rows = self._buf16("gat16", (M, self.gat_width), dev)
                    _, aij = gat_forward_rows(comp16.view(B, N, G), self.S, layer, out=gat_rows, want_attention=want_att,
                                              csr=rt.csr)
                elif CsrStructure.supported(B, N):
                    _, aij = gat_forward_rows(comp16.view(B, N, G), self.S, layer, out=gat, want_attention=want_att, csr=rt.csr)
                else:
                    gat16 = self._buf16("gat16", (M, self.gat_width), dev)
                    _, aij = gat_forward_rows(comp16.view(B, N, G), self.S, layer, out=gat16, want_attention=want_att,
                                              csr=rt.csr)
                    nat.check(lib.magat_cast_rows(nat.ptr(gat16), nat.ptr(gat), 0, M, self.gat_width, self.gat_width,
                                                  self.gat_width, stream), "magat_cast_rows")
            else:
                _, aij = gat_forward_rows(comp.view(B, N, G), self.S, layer, out=gat, want_attention=want_att, csr=rt.csr)
            layer.aij = aij
            out = self._run_actions(rt, feat, comp, gat, gat_rows, M, dev, stream)
        return out


class DecentralPlannerNet(DecentralPlannerGATNet):
    """Drop-in for the reference's GNN-baseline model class `DecentralPlannerNet` (graphs/models/decentralplanner.py:14-398; the
    first command of scripts/train_DMap.sh:30, agents/decentralplannerlocal*.py): the same per-agent CNN encoder and
    compressMLP, ONE GraphFilterBatch layer (graphML.py:5581-5700; y = bias + sum_k (x S^k) h_k with the GSO values as edge
    weights) followed by a ReLU unless config.no_ReLU, and the action MLP.  Same constructor (config object), addGSO(S)
    (:336-353: NaN scrub in place, dist_GSO_one, full_GSO), forward(x) -> (B*N, 5) logits and state_dict layout (ConvLayers.*,
    compressMLP.0.*, GFL.0.{weight (F,1,K,G), bias (F,1)}, actionsMLP.*).  Inference runs on the same gfx950 kernels as
    DecentralPlannerGATNet: magat_encoder_forward_f32, the CSR graph-filter kernels (magat_gnn_forward_csr_f32) and the
    action-head GEMM; with autograd on, the graph layer's HIP forward / backward (_GnnTrainFunction) under torch autograd.
    config.use_dilated (decentralplanner.py:57-86, 138-162: the dilated CNNs, use_dilated_version 1 | 2) is built too (round 6)."""

    def __init__(self, config):
        nn.Module.__init__(self)
        from .graphml import GraphFilterBatch
        self.config = config
        self.S = None
        self.numAgents = config.num_agents
        self.dilated_version = 0
        self.skip = "only"
        inW = inH = config.FOV + 2
        numAction = 5
        mode = config.CNN_mode
        self.cnn_mode = mode
        if getattr(config, "use_dilated", False):
            # decentralplanner.py:57-86, 138-162: the dilated CNNs ("DCP v5.1 / v5.2") take precedence over CNN_mode
            version = int(getattr(config, "use_dilated_version", 1))
            if version not in enc.DILATED_CNN:
                raise NotImplementedError("use_dilated_version %r (the reference defines 1 and 2)" % (version,))
            spec = enc.DILATED_CNN[version]
            chans, dil = spec["chans"], spec["dil"]
            layers, w, h = [], inW, inH
            for l in range(len(chans) - 1):
                layers += [nn.Conv2d(chans[l], chans[l + 1], 3, stride=1, padding=dil[l], dilation=dil[l], bias=True),
                           nn.BatchNorm2d(chans[l + 1]), nn.ReLU(inplace=True)]
                if l in (1, 3):
                    layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
                    w, h = (w - 2) // 2 + 1, (h - 2) // 2 + 1
            self.ConvLayers = nn.Sequential(*layers)
            numFeatureMap = chans[-1] * w * h
            self.cnn_mode = "Dilated"
            self.dilated_version = version
        elif mode in ("ResNetSlim_withMLP", "ResNetLarge_withMLP"):
            body = ResNetSlim() if "Slim" in mode else ResNet()
            self.ConvLayers = nn.Sequential(body, nn.Dropout(0.2), nn.Flatten(),
                                            nn.Linear(1152, config.numInputFeatures, bias=True))
            numFeatureMap = config.numInputFeatures
        elif mode in ("ResNetSlim", "ResNetLarge"):
            body = ResNetSlim() if "Slim" in mode else ResNet()
            self.ConvLayers = nn.Sequential(body, nn.Dropout(0.2))
            numFeatureMap = 1152
        else:
            chans = [3, 32, 32, 64, 64, 128]
            layers, w, h = [], inW, inH
            for l in range(5):
                layers += [nn.Conv2d(chans[l], chans[l + 1], 3, 1, 1, bias=True), nn.BatchNorm2d(chans[l + 1]),
                           nn.ReLU(inplace=True)]
                if l % 2 == 0:
                    layers.append(nn.MaxPool2d(kernel_size=2))
                    w, h = (w - 2) // 2 + 1, (h - 2) // 2 + 1
            self.ConvLayers = nn.Sequential(*layers)
            numFeatureMap = chans[-1] * w * h
            self.cnn_mode = "Default"
        self.numFeatureMap = numFeatureMap
        nif = config.numInputFeatures
        self.compressMLP = nn.Sequential(nn.Linear(numFeatureMap, nif, bias=True), nn.ReLU(inplace=True))
        self.numFeatures2Share = nif
        self.L = 1
        self.F = [nif, nif]
        self.K = [config.nGraphFilterTaps]
        self.E = 1
        self.bias = True
        gfl = [GraphFilterBatch(self.F[0], self.F[1], self.K[0], self.E, self.bias)]
        self.no_relu = bool(getattr(config, "no_ReLU", False))
        if not self.no_relu:
            gfl.append(nn.ReLU(inplace=True))
        self.GFL = nn.Sequential(*gfl)
        self.gat_width = nif
        if config.use_dropout:
            self.actionsMLP = nn.Sequential(nn.Linear(nif, nif), nn.ReLU(inplace=True), nn.Dropout(p=0.2),
                                            nn.Linear(nif, numAction), nn.Dropout(p=0.2))
        else:
            self.actionsMLP = nn.Sequential(nn.Linear(nif, numAction))
        self.apply(weights_init)
        self._rt = _Runtime()
        self._flat, self._flat_age, self._flat_epoch = None, 0, -1
        self.step_plan = True
        self.form_agents = 0
        self._cal = None

    def addGSO(self, S):
        """decentralplanner.py:336-353: aliases the caller's tensor, scrubs NaN in place, dist_GSO_one / full_GSO."""
        assert len(S.shape) == 3
        gso_mode = {"dist_GSO_one": 1, "full_GSO": 2}.get(self.config.GSO_mode, 0)
        if S.is_cuda and S.is_contiguous() and S.dtype in (torch.float32, torch.float64) and gso_mode != 2 and S.numel() > 0:
            with torch.cuda.device(S.device):
                nat.check(nat.lib().magat_gso_prepare(nat.ptr(S), 1 if S.dtype == torch.float64 else 0, S.numel(), 1, gso_mode,
                                                      nat.current_stream(S.device)), "magat_gso_prepare")
            self.S = S.unsqueeze(1)
            return
        self.S = S.unsqueeze(1)
        self.S[torch.isnan(self.S)] = 0
        if gso_mode == 1:
            self.S[self.S > 0] = 1
        elif gso_mode == 2:
            self.S = torch.ones_like(self.S).to(self.config.device)

    def returnAttentionGSO(self):
        raise AttributeError("DecentralPlannerNet has no attention (GraphFilterBatch)")

    def range_status(self):
        out = {"encoder_rerun": False, "encoder_reruns": 0, "gat_rerun": False, "gat_reruns": 0,
               "act_scales": None if self._rt is None else self._rt.act_scales}
        st = (ctypes.c_int32 * 2)()
        rt = self._rt
        if rt is not None and rt.ws is not None:
            with torch.cuda.device(rt.ws.device):
                nat.check(nat.lib().magat_encoder_read_status(nat.ptr(rt.ws), st, nat.current_stream(rt.ws.device)),
                          "magat_encoder_read_status")
            out["encoder_rerun"], out["encoder_reruns"] = bool(st[0]), int(st[1])
        return out

    def forward(self, inputTensor):
        (B, N, C, W, H) = inputTensor.shape
        dev = torch.device(self.config.device)
        x = inputTensor.reshape(B * N, C, W, H).to(dev)
        if self.S is None:
            raise TypeError("addGSO must be called before forward")
        side = self.config.FOV + 2
        if (C, W, H) != (3, side, side):
            raise RuntimeError("DecentralPlannerNet built for (3, %d, %d) state maps (config.FOV + 2), got (%d, %d, %d)"
                               % (side, side, C, W, H))
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if needs_grad or self.training:
            nat.require_device_or_composite(x, "DecentralPlannerNet in training / autograd mode")
            return self._forward_autograd(x, B, N)
        return self._forward_hip(x, B, N)

    def _forward_autograd(self, x, B, N):
        feat = convlayers_forward(self.ConvLayers, x)       # (ResNet trunks: HIP convolution kernels, train_cnn.py)
        feat = feat.view(feat.size(0), -1)
        comp = self.compressMLP(feat)
        xg = comp.reshape(B, N, self.numFeatures2Share).permute(0, 2, 1)
        self.GFL[0].addGSO(self.S)
        shared = self.GFL(xg)
        shared = shared.permute(0, 2, 1).reshape(B * N, shared.shape[1])
        return self.actionsMLP(shared)

    @torch.no_grad()
    def _forward_hip(self, x, B, N):
        if not x.is_cuda:
            raise nat.MagatNativeError("inference runs on the HIP path only; config.device=%r is not a GPU "
                                       "(no CPU fallback)" % (self.config.device,))
        dev = x.device
        M = B * N
        rt = self._refresh(dev)
        x = x.contiguous().float()
        with torch.cuda.device(dev):
            stream = nat.current_stream(dev)
            feat, comp = self._run_encoder(rt, x, M, dev, stream)
            layer = self.GFL[0]
            layer.addGSO(self.S)
            if self.S.shape[0] != B or self.S.shape[-1] < N:
                raise RuntimeError("DecentralPlannerNet: GSO of shape %s does not match a batch of %d instances x %d agents"
                                   % (tuple(self.S.shape), B, N))
            xg = comp.view(B, N, self.numFeatures2Share).permute(0, 2, 1)
            if self.S.shape[-1] == N:
                # GraphFilterBatch on the CSR kernels: rows in, rows out ((B, F, N) is a view of the (M, F) result)
                y = layer._forward_hip(xg)[0]
            else:
                # more GSO nodes than agents: the layer's own forward zero-pads the signal to the GSO's size and trims its
                # output (graphML.py:5670-5689) - the CSR structure is built for the GSO's N, so the rows must match it
                y = layer(xg)
            rows = y.permute(0, 2, 1).reshape(M, self.gat_width)
            if not self.no_relu:
                rows = torch.relu_(rows)
            return self._run_actions(rt, feat, comp, rows, rows, M, dev, stream)
