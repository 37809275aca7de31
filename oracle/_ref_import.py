"""TEST INFRASTRUCTURE ONLY -- imports the *real* reference (read-only tree at
/root/reference) inside the build container so that oracle/magat_oracle.py can
be pinned against it and golden vectors can be generated (oracle/make_golden.py).

/root/reference does not exist on the GPU box; nothing under tests -m gpu,
__graft_entry__.smoke() or bench.py may import this file.

Recipe follows SURVEY.md section 8(c): the reference's utils/__init__.py and
graphs/__init__.py eagerly import modules whose dependencies (seaborn, easydict,
tensorboardX, hashids, skimage, torchsummaryX) are not installed, so the bare
packages are pre-registered and torchsummaryX is stubbed.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("MAGAT_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "utils", "graphUtils"))


def import_reference():
    """Returns (graphML module, {bottleneckMode: DecentralPlannerGATNet class})."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    sys.dont_write_bytecode = True
    os.environ.setdefault("MPLBACKEND", "Agg")
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    for name, rel in (("utils", "utils"), ("utils.graphUtils", "utils/graphUtils"),
                      ("graphs", "graphs"), ("graphs.models", "graphs/models")):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REF_ROOT, rel)]
            sys.modules[name] = m
    if "torchsummaryX" not in sys.modules:
        stub = types.ModuleType("torchsummaryX")
        stub.summary = lambda *a, **k: None
        sys.modules["torchsummaryX"] = stub
    import importlib
    gml = importlib.import_module("utils.graphUtils.graphML")
    classes = {}
    for mode, mod in (("BottomNeck_only", "decentralplanner_GAT_bottleneck"),
                      ("BottomNeck_skipConcat", "decentralplanner_GAT_bottleneck_SkipConcat"),
                      ("BottomNeck_skipConcatGNN", "decentralplanner_GAT_bottleneck_SkipConcatGNN"),
                      ("BottomNeck_skipAddGNN", "decentralplanner_GAT_bottleneck_SkipAddGNN"),
                      ("", "decentralplanner_GAT")):
        classes[mode] = importlib.import_module("graphs.models." + mod).DecentralPlannerGATNet
    return gml, classes


def import_reference_gnn_model():
    """The GNN-baseline model class DecentralPlannerNet (graphs/models/decentralplanner.py)."""
    import_reference()
    import importlib
    return importlib.import_module("graphs.models.decentralplanner").DecentralPlannerNet


def make_config(**kw):
    """config object with the fields DecentralPlannerGATNet reads."""
    base = dict(num_agents=10, FOV=9, bottleneckFeature=128, numInputFeatures=128,
                nGraphFilterTaps=2, nAttentionHeads=1, use_dropout=False,
                CNN_mode="ResNetLarge_withMLP", attentionMode="KeyQuery",
                AttentionConcat=True, GSO_mode="dist_GSO", device="cpu",
                bottleneckMode="BottomNeck_only", batch_numAgent=True)
    base.update(kw)
    return types.SimpleNamespace(**base)
