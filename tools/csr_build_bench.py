"""Times the two halves of magat_gso_csr_build at a given shape (default BASELINE config 5: 128 x 1000 x 1000 float32):
phase 1 = the streaming pass over S (scrub + bit matrix + totals), phase 2 = the structure kernel.  MAGAT_LIB_PATH picks a build."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magat_pathplanning_amd import _native as nat
from magat_pathplanning_amd.graphml import CsrStructure
from magat_pathplanning_amd.synthetic import comm_gso
B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 1000)
dev = torch.device("cuda:0")
S = comm_gso(B, N, int(6.5 * N ** 0.5), seed=1).to(dev)
st = CsrStructure()
import os
os.environ["MAGAT_CSR_SIDE"] = "0"
st.build(S, 0, scrub_nan=1)
nnz = st.ready(dev)
lib = nat.lib()
def phase(ph):
    nat.check(lib.magat_gso_csr_build_phase(nat.ptr(S), 0, 1, 0, 0, nat.ptr(st.rowptr), nat.ptr(st.colidx), nat.ptr(st.cscptr),
                                            nat.ptr(st.csc[0]), nat.ptr(st.csc[1]), st.cap, nat.ptr(st.nnz_dev), nat.ptr(st.ws),
                                            st.ws.numel(), B, N, ph, nat.current_stream(dev)), "build")
for ph in (1, 2):
    for _ in range(3):
        phase(ph)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        phase(ph)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print("phase %d: %.1f us%s" % (ph, us, "  (%.2f TB/s over S)" % (S.numel() * 4 / us / 1e6) if ph == 1 else ""), "nnz", nnz)
