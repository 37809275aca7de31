/* magat_hip.h -- C ABI of libmagat_hip.so (MI355X / gfx950 only).
 *
 * The reference (proroklab/magat_pathplanning) is pure Python/PyTorch and has no FFI of its
 * own, so the "binding a maintainer would add" is a ctypes stub.  Every
 * entry point below names the reference code it replaces.  Conventions:
 *   - all pointers are DEVICE pointers unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call only
 *     enqueues work on that stream, never allocates, never synchronises;
 *   - the caller owns every buffer including `workspace` (size from the matching
 *     *_workspace_bytes query; 256-byte aligned);
 *   - return value: MAGAT_OK (0) or a negative MAGAT_ERR_* code; nothing throws.
 * Layout vocabulary: B planning instances, N agents per instance, M = B*N agent rows,
 * G in-features, F out-features per head, K filter taps, P attention heads.
 */
#ifndef MAGAT_HIP_H
#define MAGAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAGAT_OK 0
#define MAGAT_ERR_BAD_SHAPE (-1)   /* non-positive or inconsistent dimensions                */
#define MAGAT_ERR_UNSUPPORTED (-2) /* feature width / mode the gfx950 kernels do not cover   */
#define MAGAT_ERR_WORKSPACE (-3)   /* workspace NULL, misaligned or smaller than required    */
#define MAGAT_ERR_LAUNCH (-4)      /* hipLaunchKernel / hipGetLastError reported a failure   */
#define MAGAT_ERR_NULL (-5)        /* a required pointer is NULL                             */

#define MAGAT_MODE_KEYQUERY 0     /* attentionMode == 'KeyQuery'      graphML.py:1180-1286 */
#define MAGAT_MODE_GAT_MODIFIED 1 /* attentionMode == 'GAT_modified'  graphML.py:713-823   */
#define MAGAT_MODE_GAT_ORIGIN 2   /* attentionMode == 'GAT_origin' (GraphFilterBatchAttentional_Origin, graphML.py:4175-4339,
                                     964-1069, 1939-2002): self-loops added to the GSO (mask = |float(S)+I| > 1e-9), scores
                                     lrelu(a1.Wx_j + a2.Wx_i) without weight_bias, filter taps h[p,f,k,g] = filterWeight[0,k] * W[p,g,f]
                                     (transposed: the reference's permute+reshape, graphML.py:1967-1969).
                                     pack_weights: `taps` = filterWeight (E=1,K), `weight_bias` ignored. */
#define MAGAT_MODE_GNN 3          /* no attention: GraphFilterBatch / BatchLSIGF (graphML.py:5485-5700), the GNN baseline:
                                     z_k = z_{k-1} @ float(S), the edge weights are the GSO VALUES (CSR entry point only,
                                     magat_gnn_forward_csr_f32; P = 1, taps = weight (F,1,K,G), no ReLU inside the layer) */

int magat_abi_version(void);
/* 0 = release build.  1 = EXPERIMENT build: at least one source was compiled with timing-experiment switches (*_WHATIF_*:
 * phases of a kernel replaced by register sinks - the results are WRONG by design; tools/whatif_*.sh).  Such a source only
 * compiles under -DMAGAT_EXPERIMENT_BUILD, which also marks the library here; the Python binding refuses to load a library
 * whose flavor is not 0 unless MAGAT_ALLOW_EXPERIMENT_BUILD=1 (the probe scripts set it). (ABI 4) */
int magat_build_flavor(void);
const char* magat_error_string(int code);

/* Options.  Every tunable of the library lives in one table that is seeded from the environment (MAGAT_<NAME>) ONCE, at
 * first use, and is read / changed through these calls afterwards; nothing on the launch path calls getenv.  `name` with
 * or without the MAGAT_ prefix.  All twenty (round 5: the A/B switches of kernel forms that lost their measurements are gone
 * with those forms; round 6: + CSR_FUSED, LAT_AGENTS, GAT_WIDE_FROM; csrc/options.hip holds the table):
 *   LAT_AGENTS (512) largest agent count (magat_encoder_desc.form_agents when set) whose encoder runs ONE AGENT PER WORKGROUP
 *                    (csrc/block_lat.hip: layer1.conv2 .. layer3, pool, head and compressMLP in one launch; the batch-1 step of the
 *                    reference's inference loop); results bit-identical to the eight-agent-group kernels; 0 = never
 *   RANGE_GUARD (1)  split-arithmetic range guard (encoder and graph layer): see magat_encoder_read_status
 *   CONV_SPLIT  (7)  bit l: BasicBlock l+1 on the f16x3 split kernels; 0 = every convolution on the fp32 MFMA kernel (strict float32)
 *   CONV_PCHAIN (1), CONV_TM (2)  activation layout / tile height of the f16x3 split GEMMs (f16 plane granules against
 *                    float32 tiles, 256- against 128-agent tiles)
 *   L1_FUSED    (2)  stem + layer1.conv1 as one kernel: 2 = eight-agent groups (11 x 11 maps), 1 = row bands, 0 = two launches
 *   BLOCK_FUSED (2)  BasicBlock chain with the 6 x 6 maps of eight agents in LDS: 2 = layer1.conv2 -> layer2 -> layer3 -> pool as
 *                    ONE launch, 1 = two launches, 0 = one launch per convolution
 *   HEAD_F16    (1)  encoder head (and compressMLP) as f16x3 split products when its input is the chain kernel's pooled map
 *   HEAD_COMPRESS (1) compressMLP in the head GEMM's epilogue (one launch, bit-identical; from 32 768 agents on)
 *   HEAD_SPLITK (5120) largest agent count whose encoder head sums per-cell partials; decided on magat_encoder_desc.form_agents
 *                    when a shard sets it (bit-exact resharding with default options)
 *   CONV_BNFILL (256) f16x3 GEMMs with few agent tiles (the head at a few thousand agents) narrow their 128-column tile to 64 / 32
 *                    until the launch has this many workgroups; results are bit-identical for every value
 *   ENC_CHUNK (65536), GAT_CHUNK_MB (2048)  workspace bounds: agents per encoder pass, size of the two-launch graph layer's maps
 *   GAT_MFMA    (1)  graph layer with G = F = 128, N <= 102, K = 2 | 3, A_opt == NULL as ONE launch of matrix-core products
 *                    (maps, scores, softmax, hops; csrc/gat_mfma.hip; G = F in {32, 64}: N <= 32 csrc/gat_small.hip, 33 <= N <= 128
 *                    csrc/gat_mid.hip - round 6; G = F = 128 from GAT_WIDE_FROM (103, the lowest) to 128 agents: csrc/gat_mid.hip
 *                    with the X fragments in registers - between 102 and a larger value the two-launch form / CSR kernels); 0 = maps
 *                    GEMM + graph kernel;  GAT_SPLIT (1) that GEMM on the split kernels;  GAT_PACK (1) four instances per pass
 *                    at N <= 32 once the batch fills the chip (bit-identical)
 *   CSR_TILED   (3)  CSR path (N > 128 / bf16 storage): LDS-tiled score / hop kernels;  SKINNY (1) the action head as streamed
 *                    dot products
 *   CSR_FUSED   (1)  bf16-storage CSR layer, KeyQuery, K = 2, G = F = 128, concat, P in {1, 2, 4} (BASELINE config 5): the maps on
 *                    the matrix cores INSIDE the score / hop kernels, hop on the node features (csrc/gat_csr_fused.hip: no maps
 *                    GEMM, no Z in memory); 0 = maps GEMM + the CSR_TILED kernels
 * Returns MAGAT_ERR_UNSUPPORTED for an unknown name. */
int magat_set_option(const char* name, int value);
int magat_get_option(const char* name, int* value);
int magat_reset_option(const char* name);   /* back to the value the process started with: MAGAT_<NAME> from the environment if it
                                                was set, else the built-in default (ABI 6; was: always the built-in default) */

/* ------------------------------------------------------------------------------------------
 * GAT layer: GraphFilterBatchAttentional.forward  (utils/graphUtils/graphML.py:4636-4671)
 *   = graphAttentionLSIGFBatch_{KeyQuery,modified} (graphML.py:1724-1827)
 *   + learnAttentionGSOBatch{_KeyQuery,}           (graphML.py:1180-1286, 713-823)
 *   + ReLU / head concat or head mean              (graphML.py:4654-4667)
 *
 * X  [B,N,G]  row-major node features (= reference x (B,G,N) transposed; it is what
 *             compressMLP produces before the reference's permute, …bottleneck.py:302-306)
 * S  [B,N,N]  GSO exactly as handed to addGSO (float32, or float64 when s_is_f64 != 0);
 *             only |S| > 1e-9 is used (graphML.py:1274-1276).  NaN entries are non-edges.
 * weight      KeyQuery: (P,1,G,G); GAT_modified: (P,1,F,G)          GFL.0.weight
 * weight_bias (P,1,F)   used by GAT_modified only                   GFL.0.weight_bias
 * mixer       (P,1,2F)  used by GAT_modified only                   GFL.0.mixer
 * taps        (P,F,1,K,G)                                           GFL.0.filterWeight
 * bias        (F) or NULL                                           GFL.0.bias
 * Y  concat: [B,N,P*F] with feature index p*F+f (graphML.py:4657-4662); mean: [B,N,F]
 *    (= reference output (B,PF|F,N) transposed, i.e. already in the (B*N, features) layout
 *    actionsMLP consumes, …bottleneck.py:331).  ldy = row stride of Y in floats (>= width),
 *    so Y may be a column block of a wider skip-concat buffer.
 * A_opt [B,P,N,N] attention (aij of graphML.py:4650) or NULL (not materialised).
 * Supported: G,F in {16,32,64,128,256}, 1 <= N <= 128 (dense mask path), K >= 1, P >= 1.
 */
int magat_gat_dense_supported(int N, int G, int F); /* 1: dense-GSO kernel covers it; 0: use the *_csr_* entry point */
/* 1 when magat_gat_forward_{packed,planned}_f32 with A_opt == NULL runs this shape as ONE launch of matrix-core products
 * (gat_mfma.hip; profiling tag MAGAT_TAG_GAT_LAYER) under the current options, 0 when it takes the two-launch form (maps GEMM
 * + graph kernel).  Same results either way; tests use it to assert which kernel produced the numbers they compared. */
int magat_gat_one_launch_supported(int N, int G, int F, int K, int mode, int concat);
size_t magat_gat_packed_floats(int G, int F, int K, int P, int mode);
int magat_gat_pack_weights(const float* weight, const float* weight_bias, const float* mixer,
                           const float* taps, float* packed, int G, int F, int K, int P, int mode,
                           void* stream);
size_t magat_gat_workspace_bytes(int B, int N, int G, int F, int K, int P, int mode, int concat);
/* Range guard of the layer's per-agent maps (the f16x3 GEMM X @ [W_p | H_pk]^T; same scheme as magat_encoder_read_status):
 * the first 256 bytes of the dense AND the csr_f32 workspaces are the status block, int32 [0] = 1 when the last forward had
 * an |X| > 65504 and the maps were recomputed on the float32 MFMA kernel in the same stream, [1] = number of such re-runs
 * since the caller zeroed the block.  Synchronises `stream`. */
int magat_gat_read_status(const void* workspace, int32_t status_host[2], void* stream);
int magat_gat_forward_packed_f32(const float* X, const void* S, int s_is_f64, const float* packed,
                                 const float* bias, float* Y, int ldy, float* A_opt, void* workspace,
                                 size_t workspace_bytes, int B, int N, int G, int F, int K, int P,
                                 int mode, int concat, void* stream);

/* (ABI 7) The optional GSO plan of ABI 2-6 (magat_gat_gso_plan / magat_gat_gso_plan_bytes: edge masks + a balanced instance
 * walk made at addGSO time) is gone: opt-in, measured slower, never the default.  magat_gat_forward_planned_f32 keeps its
 * signature for existing bindings; `plan` must be NULL (anything else: MAGAT_ERR_UNSUPPORTED) - it then IS
 * magat_gat_forward_packed_f32. */
int magat_gat_forward_planned_f32(const float* X, const void* S, int s_is_f64, const float* packed,
                                 const float* bias, float* Y, int ldy, float* A_opt, void* workspace,
                                 size_t workspace_bytes, int B, int N, int G, int F, int K, int P,
                                 int mode, int concat, const void* plan, void* stream);
/* convenience: pack (into the tail of workspace) + forward; workspace must hold
 * magat_gat_workspace_bytes(...) + 4*magat_gat_packed_floats(...) bytes. */
int magat_gat_forward_dense_f32(const float* X, const void* S, int s_is_f64, const float* weight,
                                const float* weight_bias, const float* mixer, const float* taps,
                                const float* bias, float* Y, int ldy, float* A_opt, void* workspace,
                                size_t workspace_bytes, int B, int N, int G, int F, int K, int P,
                                int mode, int concat, void* stream);

/* Sparse / large-graph form of the same layer (BASELINE config 5: N = 1000 agents, comm-radius graph).  The GSO is
 * given as its edge structure: rowptr [B*(N+1)] ABSOLUTE offsets into colidx (rowptr[b*(N+1)+N] == rowptr[(b+1)*(N+1)]),
 * colidx[e] = j of the e-th edge i -> j, i.e. exactly the entries with |S[b,i,j]| > 1e-9 (graphML.py:1274-1276),
 * ascending j inside a row.  att_opt [P][nnz] receives the attention values in CSR order (or NULL).
 * Any N (<= 8190), G == F in {16,32,64,128,256}.  Gathers feature rows from global memory (L2) instead of LDS. */
size_t magat_gat_csr_workspace_bytes(int B, int N, long long nnz, int G, int F, int K, int P, int mode, int concat);
int magat_gat_forward_csr_f32(const float* X, const int* rowptr, const int* colidx, long long nnz, const float* packed,
                              const float* bias, float* Y, int ldy, float* att_opt, void* workspace,
                              size_t workspace_bytes, int B, int N, int G, int F, int K, int P, int mode, int concat,
                              void* stream);
/* GraphFilterBatch.forward (graphML.py:5670-5689 -> BatchLSIGF :5485-5579), the non-attentional GNN baseline of the paper
 * (SURVEY.md 8(f) row 2):  Y[n,f] = bias[f] + sum_k sum_g (x S^k)[g,n] h[f,0,k,g]  evaluated in the same Horner form on
 * the CSR kernels, the edge weights being the GSO values vals[e] = float(S[b,i,j]) in CSR order (edges = entries with
 * float(S) != 0: magat_gso_row_degrees / magat_gso_fill_csr with rule 2).  packed = magat_gat_pack_weights(weight = NULL,
 * NULL, NULL, taps = weight (F,1,K,G), ..., P = 1, MAGAT_MODE_GNN).  F in {16,32,64,128,256}, G % 4 == 0, any N <= 8190.
 * Workspace: magat_gat_csr_workspace_bytes(B, N, nnz, G, F, K, 1, MAGAT_MODE_GNN, 1). */
int magat_gnn_forward_csr_f32(const float* X, const int* rowptr, const int* colidx, const float* vals, long long nnz,
                              const float* packed, const float* bias, float* Y, int ldy, void* workspace,
                              size_t workspace_bytes, int B, int N, int G, int F, int K, void* stream);
/* bf16-STORAGE variant (BASELINE config 5: "1000 agents, CSR, bf16"; SURVEY.md 8(b) `..._csr_{f32,bf16}`): X, the hoisted
 * maps Z, the hop intermediates and Y are bf16 in HBM (raw uint16 bit patterns, RNE), all arithmetic accumulates in
 * fp32 (bf16 MFMA for the maps GEMM with the RNE-bf16 plane of the packed weights; fp32 scores / softmax / gathers),
 * attention values stay fp32.  Same packed weights as the f32 entry point.  Needs G % 32 == 0.  ldy in elements. */
size_t magat_gat_csr_bf16_workspace_bytes(int B, int N, long long nnz, int G, int F, int K, int P, int mode, int concat);
int magat_gat_forward_csr_bf16(const uint16_t* X, const int* rowptr, const int* colidx, long long nnz,
                               const float* packed, const float* bias, uint16_t* Y, int ldy, float* att_opt,
                               void* workspace, size_t workspace_bytes, int B, int N, int G, int F, int K, int P,
                               int mode, int concat, void* stream);
/* Row-block casts around the bf16-storage layer: src [M][ld_src] -> dst [M][ld_dst], `width` columns (multiples of 4);
 * to_bf16 != 0: float32 -> bf16 bits (RNE), else bf16 bits -> float32. */
int magat_cast_rows(const void* src, void* dst, int to_bf16, long long M, int width, int ld_src, int ld_dst, void* stream);
/* Training support (caller: loss.backward() at agents/decentralplannerlocal_OnlineExpert_GAT.py:564).
 * Forward that keeps what the backward needs: Ypre [M][P*F] = per-head filter outputs + bias BEFORE ReLU / head merge
 * (the caller's autograd owns those), att [P][nnz] (CSR order), Z [M][NC], T [(K-2)][M][P*F] (intermediate hop
 * states, NULL if K <= 2) and the CSC view (cscptr [B*(N+1)], cscsrc/cscpos/csctmp [nnz]).
 * Backward of the graph part: dZ [M][NC] (gradient wrt every column of Z) and dXd [M][G] (direct score term of dX);
 * the caller finishes with two plain GEMMs  dX = dXd + dZ @ Bt,  dBt = dZ^T @ X.  datt [P][nnz] is scratch. */
int magat_gat_train_forward_f32(const float* X, const int* rowptr, const int* colidx, long long nnz, const float* packed,
                                const float* bias, float* Ypre, float* att, float* Z, float* T, int* cscptr,
                                int* cscsrc, int* cscpos, int* csctmp, int B, int N, int G, int F, int K, int P,
                                int mode, void* stream);
int magat_gat_train_backward_f32(const float* dYpre, const float* X, const float* Z, const float* att, const float* T,
                                 const int* rowptr, const int* colidx, const int* cscptr, const int* cscsrc,
                                 const int* cscpos, long long nnz, float* dZ, float* dXd, float* datt, int B, int N,
                                 int G, int F, int K, int P, int mode, void* stream);

/* Backward of the non-attentional graph filter (GraphFilterBatch / BatchLSIGF, graphML.py:5485-5700).  The layer is linear in
 * x and in the taps:  Y = b + sum_k A^k X H_k^T  (A = row operator of "x @ S"), so  dU_k = (A^T)^k dY  is the forward hop
 * run over the CSR rows of S.  dY [M][F] -> dZ [M][K*F], slice k = dU_k.  The caller finishes with two plain GEMMs:
 * dX = dZ @ Bt (Bt [K*F][G], row k*F+f = weight[f,0,k,:]),  dweight[f,0,k,:] = (dZ^T X)[k*F+f],  dbias = column sums of dY.
 * rowptr / colidx / vals: the CSR arrays magat_gnn_forward_csr_f32 takes. */
int magat_gnn_backward_csr_f32(const float* dY, const int* rowptr, const int* colidx, const float* vals, long long nnz,
                               float* dZ, int B, int N, int F, int K, void* stream);

/* ---- Batched on-device simulator front-end: the two per-step host loops in front of the model.
 * magat_sim_gso: multiRobotSimNew.computeAdjacencyMatrix, fixed-radius branch (utils/new_simulator.py:783-804, called by
 *   getGSO :301-321): pos [B][N][2] int32 (row, col) -> S [B][N][N] float32|float64,  W = (euclidean distance < R) with zero
 *   diagonal, optional D^-1/2 W D^-1/2 (config.symmetric_norm), then W / lambda_max(W) when `normalize` (an edgeless
 *   instance stays zero).  Edge structure is bit-exact; lambda_max (-> lambda_out [B], may be NULL) comes from Lanczos +
 *   Sturm bisection in float64 (the reference: numpy.linalg.eigvalsh on the host), values agree to ~1e-12 relative.
 *   magat_sim_gso_radii: the same with one radius per instance (radii [B] float64 on the device).
 * magat_sim_connect_radius: the step-0 branch of computeAdjacencyMatrix (:759-768): r = R / 1.1, then r *= 1.1 until the
 *   graph (distance < r) is connected - the radius the episode keeps.  radii_out [B] float64 (the same float64 products as
 *   the reference, bit-exact), steps_out [B] (may be NULL) = number of growth steps, negative if still disconnected after
 *   max_steps.  Connectivity is tested by reachability from agent 0 (the reference: multiplicity of the Laplacian's zero
 *   eigenvalue, graphTools.isConnected :562-589 - the same predicate).  N <= 5461.
 * magat_sim_fov_states: AgentState.toInputTensor for guidance 'Project_G' (dataloader/statetransformer_Guidance.py:
 *   88-124, 185-239): obstacle map [B or 1][H][W] uint8 (non-zero = obstacle; outside the map counts as obstacle),
 *   pos / goal [B][N][2] int32 -> x [B][N][3][FOV+2][FOV+2] float32 in {0,1}: channel 0 obstacles, 1 goal or projected
 *   goal, 2 agents (self included); bit-exact. */
int magat_sim_gso(const int32_t* pos, double comm_radius, int symmetric_norm, int normalize, void* S, int s_is_f64,
                  double* lambda_out, int B, int N, void* stream);
int magat_sim_gso_radii(const int32_t* pos, const double* radii, int symmetric_norm, int normalize, void* S, int s_is_f64,
                        double* lambda_out, int B, int N, void* stream);
int magat_sim_connect_radius(const int32_t* pos, double comm_radius, double* radii_out, int32_t* steps_out, int B, int N,
                             int max_steps, void* stream);
int magat_sim_fov_states(const uint8_t* map, int map_batched, int H, int W, const int32_t* pos, const int32_t* goal,
                         float* x, int FOV, int B, int N, void* stream);

/* magat_sim_move: multiRobotSimNew.move + check_collision (utils/new_simulator.py:334-454, 471-520),
 *   batched: action key = argmax of the 5 logits (convectToActionKey_softmax :863-869; or given keys `actions_in`), proposed
 *   move (up 0, left 1, down 2, right 3, stop 4), shielding in the reference's order - out of the arena, face-to-face
 *   swap, obstacle, several agents into one cell, backward cascade - and pos += move in place.  ONE deviation: where the
 *   reference lets random.choice pick among several MOVING claimants of a cell, the lowest agent index wins (equal to
 *   the reference run with random.choice := first; a stationary claimant always wins, as in the reference).
 *   Outputs (each may be NULL): actions_out [B*N], moves_out [B][N][2] int8, reached_out [B][N] (new pos == goal),
 *   flags_out [B] bit0 out-of-arena, bit1 swap, bit2 obstacle, bit3 cell conflict, bit4 a position outside the map (that
 *   agent is left alone).  One workgroup per instance,
 *   H*W*4 + 16 N bytes of LDS <= 160 KB, N <= 65535. */
int magat_sim_move(const float* logits, const int32_t* actions_in, const uint8_t* map, int map_batched, int H, int W,
                   int32_t* pos, const int32_t* goal, int32_t* actions_out, int8_t* moves_out, uint8_t* reached_out,
                   int32_t* flags_out, int B, int N, void* stream);

/* magat_sim_step: one full multiRobotSimNew.move (utils/new_simulator.py:471-549) per instance with the episode state on
 * the device.  On top of magat_sim_move:
 *   - policy 0 = convectToActionKey_softmax (argmax), 1 = convectToActionKey_sum_multinorm, 2 = convectToActionKey_
 *     exp_multinorm (:863-883).  The multinomial draw is defined by caller-supplied uniforms [B*N] float64 in [0, 1)
 *     (e.g. torch.rand on the device, seeded): inverse CDF over the float32 weights (x / sum(x), or exp(x)) in index
 *     order, accumulated in float64 - the first k with w_0 + .. + w_k > u * sum.  torch.multinomial's own stream cannot be
 *     replayed on the device; the reference run with torch.multinomial patched to this rule gives identical keys
 *     (tests/golden/sim_episode_*.npz).  A row that torch.multinomial would reject (negative / inf / nan / zero sum)
 *     falls back to the argmax key and sets flag bit 5.
 *   - bookkeeping, all int32 / uint8 and updated in place: reach_goal [B][N] (sticky), first_move [B][N] (first step with a
 *     non-stop PROPOSED key, with the reference's own "== 0 means unset" test), end_step [B][N]; the step is skipped when
 *     every agent had arrived before the call or currentstep >= maxstep, and then end_step == 0 entries become
 *     currentstep - 1 and flowtime_out / makespan_out [B] are written (:528-547).  done_out [B] = allReachGoal (evaluated
 *     before the move, as the reference returns it).  flags_out != 0 (bits 0-3) is the reference's check_predictCollsion.
 *   - flag bit 4: an agent position outside the map (the agent is left where it is and takes no part in the step). */
typedef struct magat_sim_step_desc {
  const float* logits;       /* [B][N][5] or NULL */
  const int32_t* actions_in; /* [B][N] keys when logits is NULL */
  const uint8_t* map;        /* [B or 1][H][W] */
  int32_t map_batched, H, W, B, N;
  int32_t policy;            /* 0 argmax, 1 sum_multinorm, 2 exp_multinorm */
  const double* uniforms;    /* [B][N], policies 1 and 2 */
  int32_t* pos;              /* [B][N][2] in/out */
  const int32_t* goal;       /* [B][N][2] */
  uint8_t* reach_goal;       /* [B][N] in/out */
  int32_t* first_move;       /* [B][N] in/out */
  int32_t* end_step;         /* [B][N] in/out */
  int32_t currentstep, maxstep;
  int32_t* actions_out;      /* [B][N] or NULL */
  int8_t* moves_out;         /* [B][N][2] or NULL */
  int32_t* flags_out;        /* [B] or NULL */
  int32_t* done_out;         /* [B] or NULL */
  int32_t* flowtime_out;     /* [B] or NULL */
  int32_t* makespan_out;     /* [B] or NULL */
} magat_sim_step_desc;
int magat_sim_step(const magat_sim_step_desc* d, void* stream);

/* Dense GSO -> everything the CSR kernels need, for N <= 1024, in ONE streaming pass over S plus one small kernel, with no
 * host synchronisation: addGSO's in-place scrub (scrub_nan / gso_mode 0|1 as magat_gso_prepare; values are written back only
 * where they change), the edge test (edge_rule: 0 |S| > 1e-9 in S's dtype, 1 GAT_origin |float(S) + I| > 1e-9, 2 float(S) != 0)
 * -> rowptr [B*(N+1)] absolute offsets, colidx (ascending j per row), cscptr [B*(N+1)], cscsrc / cscpos (per in-edge, ascending
 * source row: source and CSR position).  cap = capacity of colidx / cscsrc / cscpos in entries: entries beyond cap are
 * DROPPED (rowptr / cscptr still hold the true offsets) and *nnz_dev, the device-side edge total, tells - the caller reads it
 * (e.g. an asynchronous copy it waits for before the graph layer, by which time the per-agent CNN is already queued), and
 * re-builds with a larger capacity when nnz > cap; B*N*N can never overflow.  The forward entry points take `nnz` as the
 * stride of their per-head attention buffers and for sizing: any value >= the true count.  Workspace: magat_gso_csr_
 * workspace_bytes (bit matrix, N*N/8 bytes per instance; 0 = N too large, use the two calls below). */
size_t magat_gso_csr_workspace_bytes(int B, int N);
int magat_gso_csr_build(void* S, int s_is_f64, int scrub_nan, int gso_mode, int edge_rule, int* rowptr, int* colidx,
                        int* cscptr, int* cscsrc, int* cscpos, long long cap, long long* nnz_dev, void* workspace,
                        size_t workspace_bytes, int B, int N, void* stream);
/* The same build in two halves, for callers that order OTHER streams behind the in-place scrub only: phase 1 = the streaming
 * pass over S (scrub, bit matrix, edge totals: the only part that writes S), phase 2 = the structure kernel (reads the
 * workspace phase 1 left, same arguments), phase 0 = both.  An event recorded between the two is all a reader of S has to
 * wait for. */
int magat_gso_csr_build_phase(void* S, int s_is_f64, int scrub_nan, int gso_mode, int edge_rule, int* rowptr, int* colidx,
                              int* cscptr, int* cscsrc, int* cscpos, long long cap, long long* nnz_dev, void* workspace,
                              size_t workspace_bytes, int B, int N, int phase, void* stream);
/* magat_gat_forward_csr_{f32,bf16} with the CSC view from magat_gso_csr_build (no per-call transpose; workspace:
 * magat_gat_csc_workspace_bytes - the csr_* figure minus the transpose scratch, 3 * nnz ints) */
size_t magat_gat_csc_workspace_bytes(int B, int N, long long nnz, int G, int F, int K, int P, int mode, int concat, int bf16);
int magat_gat_forward_csc_f32(const float* X, const int* rowptr, const int* colidx, const int* cscptr, const int* cscsrc,
                              const int* cscpos, long long nnz, const float* packed, const float* bias, float* Y, int ldy,
                              float* att_opt, void* workspace, size_t workspace_bytes, int B, int N, int G, int F, int K,
                              int P, int mode, int concat, void* stream);
int magat_gat_forward_csc_bf16(const uint16_t* X, const int* rowptr, const int* colidx, const int* cscptr,
                               const int* cscsrc, const int* cscpos, long long nnz, const float* packed, const float* bias,
                               uint16_t* Y, int ldy, float* att_opt, void* workspace, size_t workspace_bytes, int B, int N,
                               int G, int F, int K, int P, int mode, int concat, void* stream);
/* ... and with the RESULT widened to float32 by the kernel that writes it: Y [B*N][ldy] floats holding the bf16-rounded
 * values the call above stores (bit-identical after a cast), without the caller's bf16 -> float32 pass over Y */
int magat_gat_forward_csc_bf16_f32out(const uint16_t* X, const int* rowptr, const int* colidx, const int* cscptr,
                                      const int* cscsrc, const int* cscpos, long long nnz, const float* packed,
                                      const float* bias, float* Y, int ldy, float* att_opt, void* workspace,
                                      size_t workspace_bytes, int B, int N, int G, int F, int K, int P, int mode, int concat,
                                      void* stream);

/* dense GSO -> CSR in two steps (the caller prefix-sums the degrees in between): per-row edge counts, then column fill */
int magat_gso_row_degrees(const void* S, int s_is_f64, int self_loops /*edge rule: 0 |S|>1e-9, 1 GAT_origin |float(S)+I|>1e-9, 2 float(S)!=0*/, int* deg /*B*N*/, int B,
                          int N, void* stream);
int magat_gso_fill_csr(const void* S, int s_is_f64, int self_loops, const int* rowstart /*B*N*/, int* colidx, int B,
                       int N, void* stream);

/* addGSO's in-place scrub of the caller's tensor (decentralplanner_GAT_bottleneck.py:272-277):
 * scrub_nan: S[isnan(S)] = 0;  gso_mode 1 ('dist_GSO_one'): S[S>0] = 1;  2 ('full_GSO'): S = 1. */
int magat_gso_prepare(void* S, int s_is_f64, size_t count, int scrub_nan, int gso_mode, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dense per-agent maps on fp32 MFMA (v_mfma_f32_32x32x2_f32): one "segmented-K" NT GEMM that
 * serves nn.Linear, the folded conv3x3/1x1+BN(+residual)+ReLU blocks of
 * graphs/models/resnet_pytorch.py:40-73,427-524 and the GAT layer's hoisted linear maps.
 *
 * Activations are pixel-major: in[(iy*Win+ix)][m][c], m in [0,M), row stride lda floats,
 * pixel stride in_pix_stride floats.  For output pixel (oy,ox):
 *   out[(oy*Wout+ox)][m][n] = act( bias[n] + sum_{ty,tx valid} sum_c in[iy,ix][m][c] * wt[n][(ty*kW+tx)*Cin + c]
 *                                         + sum_c in2[oy*stride2, ox*stride2][m][c] * wt[n][kH*kW*Cin + c] )
 * with iy = oy*stride - pad + ty (taps falling in the zero padding are skipped, not multiplied).
 * in2 (optional, C2 > 0) is the residual branch's 1x1 strided conv or a skip-concat source (never pooled).
 * wt is [Cout][Ktot], Ktot = kH*kW*Cin + C2.  Cin, C2, lda, lda2, ldc multiples of 4.
 */
typedef struct magat_conv_gemm_desc {
  const float* in;
  const float* in2;
  const float* wt;
  const float* bias;
  float* out;
  int64_t in_pix_stride, in2_pix_stride, out_pix_stride;
  int M;
  int Cin, lda, Hin, Win, kH, kW, stride, pad, Hout, Wout;
  int C2, lda2, W2, stride2;
  int Cout, ldc, relu;
  int tag;    /* MAGAT_TAG_* used by the optional profiling hooks (0 = untagged) */
  int pool;   /* 1: `in` is read through a 2x2 SUM-pool: logical pixel (iy,ix) of the Hin x Win map =
                 sum of physical pixels (2iy+{0,1}, 2ix+{0,1}) of a map that is pool_w pixels wide
                 (AvgPool2d(2) of resnet_pytorch.py:450 with the 1/4 folded into wt);
                 2: same addressing with MAX (nn.MaxPool2d(2) of decentralplanner_GAT_bottleneck.py:137) */
  int pool_w;
  /* Operand formats.  0 = float32.  1 = "bf16x3": every value carried as three bf16 planes x1+x2+x3 (fp32-exact
   * to 2^-24), plane p at base + p * *_plane_stride elements; with in_fmt = 1 the GEMM runs on the bf16 matrix
   * cores as six partial products (conv_gemm_bf16x6.hip) and in, in2, wt are all bf16x3 (wt planes are
   * Cout*Ktot apart).  in_fmt = 2: same kernel, but in / in2 stay float32 in memory and are split into their three
   * planes by the loader on the way into LDS (wt still bf16x3).  out_fmt = 1 makes the epilogue emit the 3-plane
   * form.  in_fmt = 3: in, in2 and wt are ONE bf16 plane each (plain bf16 GEMM, fp32 accumulate); out_fmt = 2: the
   * epilogue emits one RNE bf16 plane.  in_fmt = 4 ("f16x3"): in / in2 float32, split by the loader into TWO f16 planes
   * (22 significand bits; activations beyond +-1.3e5 would saturate); wt = [2][Cout][Ktot] f16 planes of (weight * 2^e)
   * followed by one float32 2^-e (applied to the accumulator before the bias); three f16 MFMAs per product (the
   * dropped plane-2 x plane-2 term is <= 2^-22 relative) - twice the matrix-core rate of the bf16x6 form at the same
   * measured accuracy.  Formats 1-4 need Cin, C2, Cout % 32 == 0 and no pooling. */
  int in_fmt, out_fmt;
  int64_t in_plane_stride, in2_plane_stride, out_plane_stride;
  /* Agent-tile strides (elements).  Row m of a pixel lives at  (m / 128) * tile_stride + (m % 128) * ld.
   * 0 = 128 * ld, i.e. plain row-major [M][ld] per pixel (and pixel planes *_pix_stride apart: "pixel-major").
   * The encoder uses the TILE-major form [agent tile][pixel][128][C] (pix_stride = 128*C, tile_stride =
   * npix*128*C): everything a workgroup touches is one contiguous ~0.5-2 MB run instead of <= 121 pieces a
   * multi-MB plane stride apart (channel/TLB-aliasing hazard of the plane form on large batches). */
  int64_t in_tile_stride, in2_tile_stride, out_tile_stride;
  /* Granule-major agent tiles, taken by the f16x3 direct kernel only (in_fmt 4, out_fmt 0; anything else returns
   * MAGAT_ERR_UNSUPPORTED).  in_gl covers in AND in2; tiles keep their size (ld = channel count C, 128*C*4 bytes per
   * pixel and tile).
   *   1: float32 granules - inside its 128-agent tile, element (m, c) of a pixel lives at
   *      ((c / 4) * 128 + (m % 128)) * 4 + c % 4 instead of (m % 128) * ld + c: the 16-byte channel quads of the 128
   *      agents are contiguous, which is how one MFMA fragment lane per agent loads and stores them (512-byte runs per
   *      half wave).
   *   2: f16 plane granules - the tile holds the two half-precision planes of the f16x3 form (value = plane0 + plane1;
   *      plane p at byte p*256*C of the pixel's tile), each as 16-byte granules [C/8][128 agents][8 halves]; granule
   *      (T*2 + ks)*2 + h of a 32-channel tile T holds channels 32T + 16ks + 8(i>>2) + 4h + (i&3) in slot i = 0..7, i.e.
   *      exactly what MFMA lane (agent, h) of the producing epilogue holds and what the consuming loader feeds to the
   *      matrix core as its k-step-ks operand.  The CONSUMER's weights must carry the same order: within every 32-wide
   *      K slab, column 16ks + 8h + i = the weight of channel 16ks + 8(i>>2) + 4h + (i&3) (encoder.fold_resnet packs
   *      such a copy behind each f16 weight block).  Values are split once by the producer instead of once per tap by
   *      every consumer.
   *   3: as 2, but plane 1 carries, per agent and 32-channel tile, 32 bytes OCP e4m3 of the value's f16 part h1 (granules 0, 1
   *      of the tile's plane-1 region: byte 16 h + 4 g + c = channel 8 g + 4 h + c) and 32 bytes e4m3 of (value - h1) * 2^11
   *      (granules 2, 3) instead of the f16 remainder: the "f16 + MX correction" form.  The consumer (in_gl = 3, weights =
   *      the THIRD copy of the pack: plane 1 = per row and slab [e4m3(g2 * 2^5) | e4m3(g1 * 2^-6)] in the same byte order)
   *      issues, per 32-channel slab, two f16 MFMAs for h1 g1 and one v_mfma_scale_f32_32x32x64_f8f6f4 for both correction
   *      products (K block 0 = q(h1) q(g2), block 1 = q(h2) q(g1), one power-of-two scale per block).  Same bytes as layout 2,
   *      half the matrix passes; results differ from the f16x3 form by ~3e-6 of the output scale. */
  int in_gl, out_gl;
  /* Row-major float32 output split into column tiles (f16x3 direct kernel, out_gl = 0 only): when > 0, the 128-channel tile
   * t of the output lives at out + t * out_ntile_stride (+ pixel offset) with row stride ldc, instead of at column 128 t of
   * one ldc-wide row - every workgroup then writes ONE contiguous region (the GAT maps' Z: consumed as per-instance
   * [N][128] tiles).  0 = plain rows. */
  int64_t out_ntile_stride;
  /* Per-output-pixel weights (float32 kernel only): output pixel q = oy * Wout + ox uses the weight matrix at
   * wt + q * wt_pix_stride (floats), rows ldw floats apart (ldw = 0: kH*kW*Cin + C2, the packed row).  With a 1x1 kernel over
   * a pooled map this turns one long-K GEMM into Hout*Wout independent partial products written to Hout*Wout output pixels
   * - the split-K form of the encoder head for small agent counts (encoder_f32.hip sums the partials).  0 = shared weights. */
  int64_t wt_pix_stride;
  int ldw;
  /* Range guard of the split arithmetic (ABI 2).  range_flag (device int32, may be NULL): the f16x3 / f16+MX kernels OR 1
   * into it when a value left the range their half-precision planes carry exactly (|v| > 65504 on the way into a plane
   * pair; > 448 into an fp8 correction plane) - such a value was CLAMPED and the result is not fp32-class.  run_if (device
   * int32, may be NULL): the float32 kernel (in_fmt 0) does nothing unless *run_if != 0 - the stream-ordered "re-run in
   * true fp32 if the split path clamped" that magat_encoder_forward_f32 / the GAT maps use. */
  int32_t* range_flag;
  const int32_t* run_if;
  /* Activation scale of the split arithmetic (ABI 3; see "Activation scales" at magat_encoder_calibrate_f32).  in_scale
   * (device float, may be NULL = 1; a stored 0 also means 1): a POWER OF TWO the f16x3 direct kernel multiplies its float32
   * activations with on the way into their two half-precision planes (in_fmt 4, in_gl 0), and divides its accumulators by
   * - exact, and it moves small-magnitude layers into the range where the second plane is a normal number.  acc_scale
   * (device float, may be NULL): replaces the 1 / weight-scale float stored behind an f16x3 weight block.  absmax (device
   * float, may be NULL): the float32 kernel (in_fmt 0) atomically maxes the largest |output| of the launch into it
   * (calibration passes; the word is a non-negative float compared as an integer). */
  const float* in_scale;
  const float* acc_scale;
  float* absmax;
  /* bit 0: `in` holds bf16 rows, bit 1: `in2` does (lda / lda2 in bf16 elements; width and stride multiples of 8): the graph layer's
   * bf16-storage result as the action head's input, read as it is instead of through a cast pass.  Taken by the
   * streamed-dot-product form of a layer with at most 8 outputs only (float32 1x1, option SKINNY); every other kernel returns
   * MAGAT_ERR_UNSUPPORTED. */
  int bf16_rows;
  /* A SECOND 1x1 layer computed in the epilogue of the first (ABI 6; f16x3 direct kernel only: in_fmt 4, out_fmt 0, out_gl 0,
   * float32 input, ONE output pixel, Cout == Cout2 == 128, so that a workgroup tile holds whole rows): out2 [M][ldc2] =
   * act2(out . wt2^T + bias2), with `out` still written.  wt2 = [2][Cout2][Cout] f16 planes of (weight * 2^e) followed by one
   * float32 2^-e (the in_fmt 4 weight block of that layer's own launch), in_scale2 as in_scale for that layer (NULL = 1).
   * compressMLP behind the encoder head: one launch instead of two, the 128-wide feature rows never re-read; bit for bit the
   * result of the two launches.  wt2 = NULL: no second layer.  Shapes the fused form does not take (narrowed column tiles of
   * a small batch, misaligned rows, ...) return MAGAT_ERR_UNSUPPORTED before anything is launched - the caller then
   * issues the two layers on their own. */
  const void* wt2;
  const float* bias2;
  float* out2;
  const float* in_scale2;
  int Cout2, ldc2, relu2;
  /* (ABI 7) with the second layer: its rows ALSO as RNE bfloat16, [M][ldc2_bf16] (multiple of 4; NULL = not written) - the
   * bf16-storage graph layer (BASELINE config 5) reads compressMLP's rows in that type: the cast pass between them is gone. */
  void* out2_bf16;
  int ldc2_bf16;
  int dilation;      /* (ABI 8) tap spacing of the kH x kW window (nn.Conv2d dilation; float32 kernel only): 0 | 1 = dense */
} magat_conv_gemm_desc;
int magat_conv_gemm_f32(const magat_conv_gemm_desc* desc_host, void* stream);

/* (ABI 9) GraphFilterBatchAttentional.forward (graphML.py:4636-4671) like magat_gat_forward_packed_f32 (no attention output), and -
 * when the launches the layer takes leave room for it - the skinny float32 layer that reads its rows inside the layer's LAST
 * launch: `tail` describes it (what magat_conv_gemm_f32 would be called with: a 1 x 1 product with five outputs over M = B N rows -
 * the planner's action head, graphs/models/decentralplanner_GAT_bottleneck.py:341-352).  *tail_done = 1: the tail's output is
 * written; 0: the caller runs magat_conv_gemm_f32(tail) itself (always a valid outcome: other shapes, other modes, many
 * instances).  Today the room is the predicated range-guard re-run of few instances (instances x heads <= 64, KeyQuery, 32 / 64 /
 * 128 features): the closed-loop step of one planning instance is THREE launches (encoder, graph layer, re-run + action head).
 * The tail's rows come from the same device code either way: bit-identical. */
int magat_gat_forward_tail_f32(const float* X, const void* S, int s_is_f64, const float* packed, const float* bias, float* Y,
                               int ldy, void* workspace, size_t workspace_bytes, int B, int N, int G, int F, int K, int P,
                               int mode, int concat, const magat_conv_gemm_desc* tail, int* tail_done, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training of the per-agent CNN (ABI 5; agents/decentralplannerlocal_OnlineExpert_GAT.py:556-567 trains the whole module, the
 * convolutions of resnet_pytorch.py:40-73, 427-524 are 96 % of a step's FLOPs).  The training-mode forward of a convolution and
 * its INPUT gradient are magat_conv_gemm_f32 calls (bias = NULL, relu = 0: the input gradient of a stride-1 convolution is the
 * convolution of dY with mirrored taps and swapped channel roles, wt' [Cin][(kH*kW) * Cout]; of a strided one the same over the
 * zero-stuffed dY).  The WEIGHT gradient is this entry:
 *     dW[co][ty*kW+tx][ci] = sum over output pixels (oy,ox) and agents m of dY[oy*Wout+ox][m][co] * X[iy*Win+ix][m][ci],
 *     iy = oy*stride - pad + ty (taps that fall into the padding are skipped).
 * x / dy pixel-major float32 as for magat_conv_gemm_f32 (row strides lda / ldc, pixel strides in floats); Cout % 32 == 0.
 * part: magat_conv_wgrad_workspace_floats() floats = [chunks][Cout][cin_w][kH][kW] partial sums over chunks of the contraction (agent ranges x groups of output pixels), each in
 * torch's own weight layout (cin_w <= Cin: the channels the weight tensor has - the stem's rows carry a fourth, padded
 * channel); *chunks_out (host int) = how many the caller has to add up (fixed order: deterministic gradients, no atomics). */
size_t magat_conv_wgrad_workspace_floats(int M, int Cin, int cin_w, int Cout, int kH, int kW, int npix /* Hout*Wout */);
int magat_conv_wgrad_f32(const float* x, long long x_pix_stride, int lda, const float* dy, long long dy_pix_stride, int ldc,
                         float* part, int* chunks_out, int M, int Cin, int cin_w, int Cout, int Hin, int Win, int Hout, int Wout,
                         int kH, int kW, int stride, int pad, void* stream);

/* nn.BatchNorm2d in TRAINING mode (batch statistics; resnet_pytorch.py:42-58) over pixel-major rows x[rows = pixels * agents][C]
 * (contiguous, C a power of two times 4, <= 256), ReLU fused on request:
 *   forward   y = [relu]((x - mean) * invstd * gamma + beta); save_mean / save_invstd [C] for the backward; running_mean /
 *             running_var (nullable) updated in place as nn.BatchNorm does (momentum = the exponential-average factor,
 *             unbiased variance)
 *   backward  dx, dgamma [C], dbeta [C] from dy (relu: masked by y > 0, y = the forward's output)
 * workspace: magat_bn_train_workspace_floats(rows, C) floats (0 = unsupported shape).  Deterministic (fixed summation order). */
size_t magat_bn_train_workspace_floats(long long rows, int C);
int magat_bn_train_forward_f32(const float* x, float* y, long long rows, int C, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps, int relu, float* save_mean,
                               float* save_invstd, float* workspace, void* stream);
int magat_bn_train_backward_f32(const float* x, const float* y, const float* dy, float* dx, long long rows, int C,
                                const float* gamma, const float* save_mean, const float* save_invstd, int relu, float* dgamma,
                                float* dbeta, float* workspace, void* stream);

/* y[M,N] = act(x[M,K] @ w[N,K]^T + b)   (torch.nn.Linear; …bottleneck.py:105,160,229) */
int magat_linear_f32(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M,
                     int N, int K, int relu, void* stream);
/* same, with a MAGAT_TAG_* for the profiling hooks */
int magat_linear_tagged_f32(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M,
                            int N, int K, int relu, int tag, void* stream);

/* First encoder layer: conv3x3(3->32, pad 1, no bias)+BN+ReLU on the (M,3,H,W) NCHW state tensor
 * (resnet_pytorch.py:439-441, 495-498) -> pixel-major [H*W][M][32].  wt [32][27] BN-folded
 * (index c*9+ty*3+tx), bias [32]. */
int magat_conv_first_f32(const float* x, const float* wt, const float* bias, float* out, int M, int H,
                         int W, void* stream);
/* same, writing the TILE-major form [agent tile of 128][H*W][128][32] (out must hold ceil(M/128) full tiles) */
int magat_conv_first_tiled_f32(const float* x, const float* wt, const float* bias, float* out, int M, int H,
                               int W, void* stream);

/* ------------------------------------------------------------------------------------------
 * Whole per-agent encoder  ConvLayers (+Flatten+Linear for *_withMLP) -> compressMLP
 * (decentralplanner_GAT_bottleneck.py:90-166, 291-302) from a BN-folded parameter pack.
 * variant: 0 = ResNet(BasicBlock,[1,1,1]) "ResNetLarge", 1 = ResNetSlim(BasicBlock,[1,1]),
 * 2 = CNN_mode "Default": 5 x [conv3x3(bias)+BN+ReLU], MaxPool2d(2) after layers 0, 2, 4
 * (decentralplanner_GAT_bottleneck.py:118-147); pack: off[0..1] conv0, off[2+2i], off[3+2i] weight/bias of conv i+1
 * (i = 0..3), off[14] 128x128 identity (final max-pool as a pooled 1x1 GEMM), off[16..17] compressMLP.
 * The pack layout is produced by magat_pathplanning_amd.encoder.fold_resnet (documented there
 * and in DESIGN.md); offsets are passed explicitly so the ABI does not hard-code it.
 */
typedef struct magat_encoder_desc {
  int variant;     /* 0 ResNetLarge, 1 ResNetSlim, 2 CNN_mode "Default"; (ABI 8) 3 / 4: the dilated CNNs of DecentralPlannerNet
                      (config.use_dilated, use_dilated_version 1 / 2; graphs/models/decentralplanner.py:57-86, 138-162): conv-BN-ReLU
                      x 5 (x 4) with dilation 1 3 1 3 (1), MaxPool2d(2) behind layers 1 and 3; pack as for variant 2 */
  int H, W;        /* FOV+2 (11) */
  int n_feat;      /* numFeatureMap: width of `feat` (128 for *_withMLP, 1152 otherwise) */
  int n_comp;      /* bottleneckFeature G (0: skip compressMLP) */
  const float* pack; /* device pointer to the folded parameter pack */
  int64_t off[32];   /* float offsets into pack: see DESIGN.md "encoder pack" */
  int64_t chain_off; /* float offset of the BasicBlock chain kernel's fragment-major weights (encoder.pack_chain_weights: layer1.
                        conv2+downsample, layer2.conv1, layer2.conv2+downsample), 0 = absent -> layer-by-layer kernels (ABI 2) */
  int64_t chain3_off; /* float offset of the layer3 kernel's weights (encoder.pack_block3_weights), 0 = absent (ABI 2) */
  int64_t head16_off; /* float offset of the head weight as f16x2 planes + 2^-e (in_fmt 4), 0 = absent: the head runs as
                         f16x3 split products when its input is the layer3 kernel's pooled map (ABI 2) */
  int64_t comp16_off; /* the same for the compressMLP weight: 0 = absent (float32 MFMA); used together with head16_off (ABI 2) */
  int64_t scaled_off; /* float offset of the ACTIVATION-SCALE block (encoder.fold_activation_scales; ABI 3), 0 = absent:
                         [w0' 864 | b0' 32 | b1' 32 | bA' 32 | bB' 64 | bC' 64 | b31' 128 | b32' 128 | sA' sB' sC' s31' s32'
                          head_in feat_in pad] - the stem weights / biases of the fused chain multiplied by the power-of-two
                         scale their layer's activations are carried with, the five 1 / weight-scale floats with the scale
                         ratios folded in, and the in_scale of the head's and compressMLP's float32 loaders */
  int64_t l1frag_off; /* float offset of layer1.conv1 as fragment-major f16 planes (encoder.pack_chain_weights(rows, 32, 0):
                         [tap 9][k step 2][plane 2] 1 KB blocks + [2^-e, 0, 0, 0]; ABI 4), 0 = absent.  With it (11 x 11 maps,
                         option L1_FUSED = 2) the stem and layer1.conv1 run as the eight-agent-group kernel of block_fused.hip
                         (every stem pixel computed once, operands read straight out of LDS) instead of layer1_fused.hip */
  int form_agents;    /* ABI 6: the agent count the batch-size-dependent kernel FORMS are chosen on (the encoder head's split-K
                         form below option HEAD_SPLITK), 0 = this call's M.  A shard of a larger batch passes the GLOBAL agent
                         count here (distributed.sharded_forward does), so that every shard sums in the order the whole batch
                         would: shards then concatenate to the single-process result bit for bit whatever their size */
  void* comp_bf16;    /* ABI 7: NULL, or [M][n_comp] bfloat16 rows that receive RNE-bf16(comp) as well: written by the epilogue that
                         produces comp where it can (compressMLP in the head's launch), by a cast pass otherwise; after a range-guard
                         re-run of the encoder they are rewritten from the float32 rows (same stream, predicated on the flag) */
  int64_t headfrag_off; /* ABI 8: float offsets of the head's and compressMLP's weights as FRAGMENT-major f16 planes (encoder.
                           pack_frag_natural: the values and scale of head16 / comp16 in the order a wave fetches them), 0 = absent.  With */
  int64_t compfrag_off; /* both (ResNetLarge, 11 x 11 maps, n_feat = n_comp = 128) the few-agent encoder runs head and compressMLP in the
                           epilogue of the one-agent-per-workgroup chain kernel (block_lat.hip; option LAT_AGENTS): two launches for the
                           whole encoder, its results bit-identical to the batched forms (long-K f16x3 head, f16x3 compressMLP) */
} magat_encoder_desc;
/* Activation scales (ABI 3).  The split arithmetic carries a value as two f16 planes: exact for |v| <= 65504, but the SECOND
 * plane is a full 11-bit number only for |v| >~ 0.25 - a layer whose activations are all small (a small BatchNorm gamma: an
 * ordinary thing in a trained checkpoint) would be carried with an absolute floor of 2^-25 per value instead of fp32's
 * relative 2^-24.  ReLU networks commute with positive scaling, so every plane-forming stage of the fused encoder path carries
 * its map multiplied by a power of two chosen from the layer's measured magnitude (target: largest value ~2^10; exact, folded
 * into biases and into the 1 / weight-scale of the epilogues host-side), and the float32 loaders of the head, compressMLP
 * and the graph layer's maps multiply on load (magat_conv_gemm_desc.in_scale).  magat_encoder_calibrate_f32 measures the
 * magnitudes: ONE float32 pass (the range guard's re-run path: layer-by-layer float32 MFMA kernels) that also writes feat /
 * comp, leaving in absmax [16] (device floats, zeroed here) the largest |output| of: [0] stem, [1 + 2 l] layer(l+1).conv1,
 * [2 + 2 l] layer(l+1).conv2 + downsample (l = 0..2), [7] feat, [8] comp.  Unscaled packs (scaled_off = 0) behave as before. */
int magat_encoder_calibrate_f32(const magat_encoder_desc* desc_host, const float* x, float* feat, int ldfeat, float* comp,
                                int ldcomp, void* workspace, size_t workspace_bytes, int M, float* absmax, void* stream);
/* The first stage of the fused encoder path on its own (ABI 4; what magat_encoder_forward_f32 launches first, exported so
 * that the stage can be tested and profiled in isolation): stem conv3x3(3 -> 32)+BN+ReLU fused with layer1.conv1
 * (32 -> 32, stride 2)+BN+ReLU.  x (M,3,H,W) -> out: layer1.conv1's output, ctr: the stem output at the stride-2 pixels (the
 * input of the block's residual 1x1 branch), both as [ceil(M/128)][Ho*Wo] f16 plane-granule tiles of 32 channels
 * (in_gl = 2 above; 128*32*4 bytes per tile, the caller provides whole tiles).  form 2: groups of eight agents, every stem pixel
 * computed once (block_fused.hip stem8_kernel; 11 x 11 maps, needs desc->l1frag_off); form 1: 64-agent row bands
 * (layer1_fused.hip); form 0: what the encoder would pick (option L1_FUSED).  Uses the activation-scale block when
 * desc->scaled_off is set.  range_flag (device int32, may be NULL) is OR-ed with 1 when a value left the planes' range. */
int magat_encoder_stem_block_f32(const magat_encoder_desc* desc_host, const float* x, void* out, void* ctr, int M, int form,
                                 int32_t* range_flag, void* stream);
/* Range guard (option RANGE_GUARD, default 1).  The convolutions run as f16x3 split products (two half-precision planes per
 * value, fp32 accumulate: as accurate as the fp32 MFMA kernel while every activation stays within +-65504; the fused stem
 * carries its output 16x and needs it below 4094).  Whether a forward stayed inside is checked ON THE DEVICE: every kernel
 * that forms planes ORs a flag when it had to clamp, and the encoder then re-runs itself on the float32 MFMA kernels in
 * the same stream, predicated on that flag (two launches that return immediately when the flag is clear: the stem, and every
 * layer behind it chained in one kernel whose last workgroup also does the flag's bookkeeping), so feat / comp are fp32-class
 * either way.  The first 256 bytes of `workspace` are the status block, which the caller zeroes ONCE,
 * when it allocates the workspace (the library keeps its working flag there, clear between forwards).
 * magat_encoder_read_status copies two words to the host: status_host[0] = 1 if the LAST forward clamped and was re-run in
 * float32, status_host[1] = number of re-run forwards since the block was zeroed; it is the one call that synchronises
 * `stream`.  Cost of the guard when nothing clamps: the two launches above plus two for the graph layer's re-run, ~20 us of a
 * 2.6 ms c3 step (measured).  The block's words [0..2] and [6] are the library's, [4] is the graph layer's input scale. */
size_t magat_encoder_workspace_bytes(const magat_encoder_desc* desc_host, int M);
int magat_encoder_read_status(const void* workspace, int32_t status_host[2], void* stream);
int magat_encoder_forward_f32(const magat_encoder_desc* desc_host, const float* x /*M,3,H,W*/,
                              float* feat, int ldfeat, float* comp, int ldcomp, void* workspace,
                              size_t workspace_bytes, int M, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optional per-kernel timing (bench.py roofline leg).  When enabled every kernel the library
 * launches is bracketed by hipEvents on its launch stream; after the caller synchronises,
 * magat_profile_collect() folds them into per-tag (count, total ms).  Off by default.
 */
#define MAGAT_TAG_UNTAGGED 0
#define MAGAT_TAG_CONV_FIRST 1
#define MAGAT_TAG_BLOCK_CONV 2  /* +2*l: conv1 of BasicBlock l, +2*l+1: conv2(+downsample) of block l (l=0..2) */
#define MAGAT_TAG_HEAD 8        /* avgpool+fc(+Linear) folded conv */
#define MAGAT_TAG_COMPRESS 9
#define MAGAT_TAG_GAT_MAPS 10   /* hoisted X @ [W_p | H_pk]^T GEMM */
#define MAGAT_TAG_GAT_GRAPH 11  /* scores + softmax + hops kernel */
#define MAGAT_TAG_ACTIONS 12
#define MAGAT_TAG_HEAD_MEAN 13
#define MAGAT_TAG_GAT_PACK 14
#define MAGAT_TAG_GSO_PREPARE 15
#define MAGAT_TAG_GAT_PREPARE 16  /* edge masks + edge counts + balanced instance order for the persistent graph kernel */
#define MAGAT_TAG_RANGE_GUARD 17  /* flag reset + the predicated float32 re-run launches of the range guard (no-ops when clear) */
#define MAGAT_TAG_BLOCK_CHAIN 18  /* BasicBlock chain kernel (block_fused.hip) */
#define MAGAT_TAG_GAT_LAYER 19    /* the KeyQuery layer as one launch of matrix-core products (gat_mfma.hip) */
#define MAGAT_TAG_GSO_CSR 20      /* dense GSO -> CSR + CSC structure (magat_gso_csr_build, or the transpose inside *_csr_*) */
#define MAGAT_TAG_GAT_CAST 21     /* float32 <-> bf16 row casts around the bf16-storage graph layer */
#define MAGAT_TAG_BLOCK3 22       /* layer3 + ReLU + 2x2 pool in one launch (block_fused.hip) */
#define MAGAT_TAG_BLOCK_FULL 23   /* layer1.conv2 -> layer2 -> layer3 -> pool in one launch (block_fused.hip) */
#define MAGAT_TAG_CONV_WGRAD 24   /* weight gradient of a convolution (conv_train.hip; training) */
#define MAGAT_PROF_TAGS 25
int magat_gat_set_debug_buffer(long long* dev_buf); /* [grid][8] int64 phase timestamps of gat_dense_kernel; NULL = off */
int magat_profile_reserve(int spans);   /* pre-create event pairs (keeps hipEventCreate out of a timed region) */
int magat_profile_enable(int on);
int magat_profile_collect(void);
int magat_profile_read(int tag, long long* count, double* total_ms);
int magat_profile_reset(void);
/* What the device SUSTAINS on v_mfma_f32_32x32x16_f16 from registers alone (operand bits toggling), one wave per SIMD on every
 * CU, in one launch of about ms_target milliseconds: *tflops (dense f16).  The clock the chip holds under matrix load is part
 * of the figure (MI355X: 1.5-1.6 PFLOP/s against the 2.5 PFLOP/s of the 2.4 GHz peak clock).  scratch: 256 * CUs floats.
 * Synchronises the stream.  (ABI 5; bench.py's `roofline.sustained_*` keys) */
int magat_mfma_sustained_f16(double* tflops, float* scratch, int ms_target, void* stream);
/* (ABI 6) the same launch with its own clock evidence: *clock_mhz = the core clock the chip held inside it (median over the CUs
 * of core-clock cycles / 100 MHz ticks, read by the kernel itself), *per_clk = flop per clock and SIMD the rate amounts to at
 * that clock (1024 = one v_mfma_f32_32x32x16_f16 issued every 32 cycles: a matrix pipe that never idles), so that
 * sustained = clock x per_clk x SIMDs.  Either may be NULL.  scratch: 256 * CUs floats + 2 * CUs int64 behind them. */
int magat_mfma_sustained_f16_ex(double* tflops, double* clock_mhz, double* per_clk, float* scratch, int ms_target, void* stream);
/* Which FORM of a kernel the launches took since magat_form_reset (host-side counters, bumped where a launcher decides): the
 * forms that exist only at benchmark sizes are asserted by the full-size parity tests (tests/test_gpu_fullsize.py). */
#define MAGAT_FORM_HEAD_LONGK 0   /* encoder head as ONE long-K f16x3 GEMM (agents above option HEAD_SPLITK) */
#define MAGAT_FORM_HEAD_SPLITK 1  /* encoder head as per-cell partial products + sum (few agents) */
#define MAGAT_FORM_GAT_PACK 2     /* one-launch graph kernel, four instances per pass (N <= 32, batch fills the chip) */
#define MAGAT_FORM_GAT_PERSIST 3  /* one-launch graph kernel, more planning instances than workgroups (persistent walk) */
#define MAGAT_FORM_GAT_HSPLIT 4   /* one-launch graph kernel, a workgroup per (instance, head) (small batches) */
#define MAGAT_FORM_CHAIN_PERSIST 5 /* chain kernel: more 8-agent groups than workgroups (persistent group loop) */
#define MAGAT_FORM_HEAD_COMPRESS 6 /* compressMLP computed in the head GEMM's epilogue (one launch for both) */
#define MAGAT_FORM_GUARD_ONE 7    /* range guard of the encoder as one predicated launch */
#define MAGAT_FORM_CSR_FUSED 8    /* bf16-storage CSR layer with the maps inside the graph kernels (gat_csr_fused.hip) */
#define MAGAT_FORM_GAT_MID 9      /* one-launch graph layer for G = F in {32, 64} on 33 .. 128 agents (gat_mid.hip) */
#define MAGAT_FORM_CHAIN_LAT 10   /* chain kernel, latency form: one agent per workgroup (block_lat.hip; option LAT_AGENTS) */
#define MAGAT_FORM_HEAD_LAT 11    /* ... with the encoder head and compressMLP in its epilogue (ABI 8: headfrag_off / compfrag_off) */
#define MAGAT_FORM_GUARD_LAT 12   /* ... and the encoder's range guard inside the same launch (no predicated launches behind it) */
#define MAGAT_FORM_STEM_LAT 13    /* ... and the stem + layer1.conv1 in front: the whole encoder of a few-agent call is ONE launch */
#define MAGAT_FORM_ACTIONS_TAIL 14 /* the action head inside the graph layer's predicated re-run launch (magat_gat_forward_tail_f32) */
#define MAGAT_FORMS 15
long long magat_form_count(int id);
int magat_form_reset(void);

#ifdef __cplusplus
}
#endif
#endif /* MAGAT_HIP_H */
