/*
 * hnrf.h -- C ABI of the MI355X-native HumanNeRF ray-marching path
 * (libhnrf.so, hand-written HIP for gfx950).
 *
 * The reference (ChenYutongTHU/humannerf) is pure Python/PyTorch and has no FFI;
 * its "operator interface" for this path is the set of methods of
 * core/nets/human_nerf/network.py::Network.  Each entry point below replaces
 * the torch-op sequence of one of those methods and cites it.  A maintainer of
 * the reference binds these with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions:
 *  - every pointer is a DEVICE pointer to contiguous row-major fp32 unless
 *    stated otherwise; the caller owns every buffer, the library allocates
 *    nothing and keeps no state between calls;
 *  - `stream` is a hipStream_t passed as void*; calls are stream-ordered and
 *    never synchronise the device or touch the host copy of any buffer;
 *  - return value 0 = success, negative = error (see HNRF_E_*); the message is
 *    available from hnrf_last_error() (thread-local); nothing throws or exits;
 *  - nullable outputs may be passed as NULL to skip their HBM writes.
 */
#ifndef HNRF_H
#define HNRF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HNRF_OK            0
#define HNRF_E_ARG        -1   /* null pointer / bad dimension            */
#define HNRF_E_UNSUPPORTED -2  /* layer shape / mode not built            */
#define HNRF_E_LAUNCH     -3   /* hipLaunch / hipGetLastError failed      */
#define HNRF_E_WORKSPACE  -4   /* workspace too small                     */

/* arithmetic of the per-sample MLP GEMMs */
#define HNRF_MLP_F32    0      /* v_mfma_f32_32x32x2_f32: exact fp32 fma chain   */
#define HNRF_MLP_F16X3  1      /* split-fp16 (hi+lo) inputs, 3 MFMAs, fp32 accum */
#define HNRF_MLP_F16X3_H 2     /* training entry points only (hnrf_*_fwd_train, hnrf_*_bwd): HNRF_MLP_F16X3 arithmetic with
                                * the saved weight-gradient operands in f16 -- acts / dZ are f16 matrices of the same
                                * layer count in the BLOCKED layout of hnrf_mlp_dw_h (every layer padded to a multiple of 128
                                * samples: [L][ceil(P / 128) * 128][width]), pe_out is row-major f16 [P][64] (zero-padded), dZ
                                * carries the chain's power-of-two scale and dz_amax receives that scale per layer ([L]
                                * floats) for hnrf_mlp_dw_h.  Packed images are the HNRF_MLP_F16X3 ones. */

int         hnrf_abi_version(void);
const char* hnrf_last_error(void);

/* ---- f16-range guard of the HNRF_MLP_F16X3 inference kernels ------------------
 * The split v = hi + lo only holds below 65504: the kernels clamp post-ReLU activations there, so a checkpoint whose
 * hidden activations leave the f16 range would render a wrong image without any error (the reference's fp32 nn.Linear
 * chains, mlp_rgb_sigma.py:163-198 / mlp_offset.py:74-84, have no such limit).  Every packed image therefore carries a
 * STATUS WORD (uint32 at byte offset hnrf_*_status_offset(mode) of the caller's `packed` buffer; zeroed by every
 * hnrf_*_pack): hnrf_canonical_fwd / hnrf_nonrigid_fwd, their sparse forms and the hnrf_render_* entries OR
 * HNRF_STATUS_F16_RANGE into it when any activation reached 6e4.  This is the one place where a forward call
 * writes into `packed`.  The caller reads the word when it likes (no call synchronises) and re-renders with
 * HNRF_MLP_F32, which has no such limit (offset 0 = the mode has no status word).  Training has its own guard on
 * the saved activations (humannerf_amd/autograd.py OperandRangeGuard). */
#define HNRF_STATUS_F16_RANGE 1u
size_t hnrf_canonical_status_offset(int mode);
size_t hnrf_nonrigid_status_offset(int mode);
/* The guard costs 5 VALU instructions per 8 activations (+2.5 % canonical-kernel time, -3.0 % rays/s on the
 * 512x512x128 frame, measured A/B).  Flags OR-ed into the `mode` argument of the forward entry points select unguarded
 * kernel instances; the low byte stays the arithmetic:
 *   (none)                       every launch is guarded (the default of every entry point);
 *   HNRF_MLP_NO_RANGE_GUARD      no launch is guarded -- the status words stay as they are;
 *   HNRF_MLP_GUARD_ONE_CHUNK     hnrf_render_frame_fwd only: of the frame's ray chunks only number
 *                                ((unsigned)mode >> 16) % n_chunks is guarded -- the caller rotates that index from
 *                                frame to frame (an audit: humannerf_amd.network guards every chunk of the first frame
 *                                after a weight change and one rotating chunk afterwards, cfg.amd.f16_range_guard). */
#define HNRF_MLP_ARITH_MASK       0xff
#define HNRF_MLP_NO_RANGE_GUARD   0x100
#define HNRF_MLP_GUARD_ONE_CHUNK  0x200

/* ---- K1: z-sampling + inverse-LBS warp ------------------------------------
 * Replaces Network._get_samples_along_ray / _stratified_sampling
 * (network.py:455-471), pts = o + d*z (network.py:499) and
 * Network._sample_motion_fields (network.py:392-444).
 *  rays_o, rays_d [R,3]; near, far [R]; t_rand [R,S] or NULL (perturb == 0);
 *  motion_Rs [B,3,3], motion_Ts [B,3]; vol [>=B, G,G,G] (background channel, if
 *  present, is not read); bbox_min, bbox_scale [3]   -- all device pointers.
 * Outputs: z_vals [R,S], x_skel [R,S,3], fg_mask [R,S] (= sum of weights,
 *  unclamped), bmw [R,S,B] or NULL (unnormalised per-bone weights; with
 *  B == 24 16-byte aligned -- it is then written in 16-byte pieces --,
 *  HNRF_E_ARG otherwise).
 * Every operation of the reference's tensor expressions is rounded on its own
 * (no compiler-chosen fma), so all forms of the kernel agree bit for bit. */
int hnrf_sample_warp_fwd(const float* rays_o, const float* rays_d,
                         const float* near, const float* far, const float* t_rand,
                         const float* motion_Rs, const float* motion_Ts,
                         const float* vol, const float* bbox_min, const float* bbox_scale,
                         int64_t R, int S, int B, int G,
                         float* z_vals, float* x_skel, float* fg_mask, float* bmw,
                         void* stream);

/* ---- sample culling (no counterpart in the reference, which evaluates every sample) --------
 * idx[0 .. *count) = indices p with fg_mask[p] >= eps, count written on the device (no host
 * sync).  alpha = (1 - exp(-sigma delta)) * fg_mask (network.py:369) < eps for a dropped sample,
 * so a ray's rgb / alpha / depth move by at most ~2 S eps; eps == 0 keeps every sample and the
 * path is exactly the reference's.  idx must hold P ints. */
int hnrf_compact_samples(const float* fg_mask, float eps, int64_t P, int* idx, int* count, void* stream);

/* ---- K2: non-rigid motion MLP ---------------------------------------------
 * Replaces hannw_fourier embed (embedders/hannw_fourier.py:21-49) +
 * NonRigidMotionMLP.forward (non_rigid_motion_mlps/mlp_offset.py:74-114) as
 * called from Network._apply_mlp_kernals (network.py:264-275).
 * Default architecture only: 6 x 128, skip [h | PE36] at layer 4, out 3,
 * condition code 69 (one vector per frame, folded into the first bias).
 *
 * hnrf_nonrigid_pack: weights[7], biases[7] are the nn.Linear tensors
 *  block_mlps.{0,2,...,12} in (out,in) layout; cond [69] is the (possibly
 *  zeroed, network.py:735-737) condition code.  Writes the MFMA-ordered weight
 *  image into `packed` (hnrf_nonrigid_packed_bytes(mode) bytes).  Re-run after
 *  every parameter update or new frame (cond changes per frame). */
size_t hnrf_nonrigid_packed_bytes(int mode);
int hnrf_nonrigid_pack(const float* const* weights, const float* const* biases,
                       const float* cond, int mode, void* packed, void* stream);
/*  x_skel [P,3]; hann_w [6] device pointer (window weights, hannw_fourier.py:
 *  26-40).  Outputs xyz = x_skel + offset [P,3]; offsets [P,3] or NULL. */
int hnrf_nonrigid_fwd(const float* x_skel, const float* hann_w, const void* packed,
                      int mode, int64_t P, float* xyz, float* offsets, void* stream);
int hnrf_nonrigid_fwd_sparse(const float* x_skel, const float* hann_w, const void* packed,
                             int mode, int64_t P, const int* idx, const int* count,
                             float* xyz, float* offsets, void* stream);

/* ---- K3: canonical MLP ----------------------------------------------------
 * Replaces fourier embed (embedders/fourier.py:9-38) + CanonicalMLP.forward
 * default branch (canonical_mlps/mlp_rgb_sigma.py:132-198) as called from
 * Network._apply_mlp_kernals (network.py:305-315).
 * Default architecture only: PE63 -> 8 x 256, skip [PE63 | h] at layer 5, out 4.
 *  weights[9], biases[9]: pts_linears.{0,...,14} then output_linear.0. */
size_t hnrf_canonical_packed_bytes(int mode);
int hnrf_canonical_pack(const float* const* weights, const float* const* biases,
                        int mode, void* packed, void* stream);
/*  xyz [P,3] -> raw [P,4] = (r,g,b,sigma) pre-activation. */
int hnrf_canonical_fwd(const float* xyz, const void* packed, int mode, int64_t P,
                       float* raw, void* stream);
/*  Sparse form: only the samples idx[0 .. *count) are evaluated (xyz read at and raw written to
 *  those indices; everything else untouched).  idx, count: device pointers from
 *  hnrf_compact_samples; P = capacity of idx (grid size). */
int hnrf_canonical_fwd_sparse(const float* xyz, const void* packed, int mode, int64_t P,
                              const int* idx, const int* count, float* raw, void* stream);

/* ---- K4: alpha compositing --------------------------------------------------
 * Replaces Network._raw2outputs (network.py:355-388).
 *  raw [R,S,4]; fg_mask [R,S]; z_vals [R,S]; rays_d [R,3]; xyz [R,S,3] (may be
 *  NULL when cnl_xyz is NULL); bgcolor [3] in 0..255 (device pointer).
 *  cull_eps: samples with fg_mask < cull_eps get weight 0 and their raw is not interpreted
 *  (0 = reference behaviour).
 * Outputs: rgb [R,3], alpha [R], depth [R]; nullable: weights [R,S],
 *  rgb_on_rays [R,S,3], cnl_xyz [R,3], cnl_rgb [R,3], cnl_weight [R]. */
int hnrf_composite_fwd(const float* raw, const float* fg_mask, const float* z_vals,
                       const float* rays_d, const float* xyz, const float* bgcolor,
                       int64_t R, int S, float cull_eps,
                       float* rgb, float* alpha, float* depth,
                       float* weights, float* rgb_on_rays,
                       float* cnl_xyz, float* cnl_rgb, float* cnl_weight,
                       void* stream);

/* ---- whole path for one ray chunk -------------------------------------------
 * Replaces Network._render_rays (network.py:474-602): K1 -> K2 -> K3 -> K4 on
 * `stream`, intermediates in caller-provided workspace
 * (hnrf_render_workspace_bytes(R,S) bytes, 256-byte aligned).
 * nr_packed == NULL means cfg.ignore_non_rigid_motions (network.py:264,276-277).
 * cull_eps > 0: the MLPs run only on the samples with fg_mask >= cull_eps (see
 * hnrf_compact_samples); 0 = every sample, as the reference.
 * Only rgb/alpha/depth are written (the trainer and the image writers read
 * nothing else: trainer.py:121, run.py:130).
 * ev_mlp_start / ev_mlp_stop: optional hipEvent_t (as void*, may be NULL) recorded
 * on `stream` right before / after the canonical-MLP launch, so a caller can time
 * the dominant kernel without a profiler (bench.py roofline). */
size_t hnrf_render_workspace_bytes(int64_t R, int S);
int hnrf_render_rays_fwd(const float* rays_o, const float* rays_d,
                         const float* near, const float* far, const float* t_rand,
                         const float* motion_Rs, const float* motion_Ts,
                         const float* vol, const float* bbox_min, const float* bbox_scale,
                         const float* hann_w, const void* nr_packed, const void* cnl_packed,
                         const float* bgcolor, int mode, float cull_eps,
                         int64_t R, int S, int B, int G,
                         void* workspace, size_t workspace_bytes,
                         float* rgb, float* alpha, float* depth,
                         void* ev_mlp_start, void* ev_mlp_stop, void* stream);

/* Opt-in variant with early ray termination (NOT the reference arithmetic): the samples are walked front to back
 * in slabs of 32; a ray whose transmittance has fallen below term_eps (0 < term_eps < 1) is not evaluated further,
 * which moves rgb / alpha by at most term_eps; cull_eps as above (may be 0).  evaluated (nullable): device int that
 * receives the number of samples that went through the MLPs.  workspace: hnrf_render_term_workspace_bytes. */
size_t hnrf_render_term_workspace_bytes(int64_t R, int S);
int hnrf_render_rays_term_fwd(const float* rays_o, const float* rays_d,
                              const float* near, const float* far, const float* t_rand,
                              const float* motion_Rs, const float* motion_Ts,
                              const float* vol, const float* bbox_min, const float* bbox_scale,
                              const float* hann_w, const void* nr_packed, const void* cnl_packed,
                              const float* bgcolor, int mode, float cull_eps, float term_eps,
                              int64_t R, int S, int B, int G,
                              void* workspace, size_t workspace_bytes,
                              float* rgb, float* alpha, float* depth, int* evaluated, void* stream);

/* Whole frame: Network._batchify_rays (network.py:330-352) over _render_rays.  N rays in chunks of `chunk` (cfg.chunk)
 * through K1..K4, rgb [N,3] / alpha [N] / depth [N] written for the whole frame.  The eight diagnostic outputs of the
 * reference's forward are optional, all or none (whole-frame buffers: weights_on_rays [N,S], rgb_on_rays [N,S,3],
 * cnl_xyz [N,3], cnl_rgb [N,3], cnl_weight [N], xyz_on_rays [N,S,3], bmw [N,S,B], offsets [N,S,3]); sample culling
 * (cull_eps > 0) only without them.  workspace: hnrf_render_frame_workspace_bytes(chunk, S), 256-byte aligned (two chunk
 * workspaces that alternate).
 * side_stream (nullable): a second stream of the same device on which the LBS warp (K1) of chunk i+1 runs while the
 * MLP kernels of chunk i occupy `stream` -- K1 has no LDS and few registers, so it shares the CUs with them.  Needs
 * events = 5 hipEvent_t owned by the caller (no timing needed); all ordering between the two streams is expressed
 * through them, nothing is synchronised with the host.  On return every result is ordered on `stream`.
 * mlp_events (nullable): 2 x ceil(N / chunk) hipEvent_t recorded around every canonical-MLP launch (for timing). */
size_t hnrf_render_frame_workspace_bytes(int64_t chunk, int S);
int hnrf_render_frame_fwd(const float* rays_o, const float* rays_d, const float* near, const float* far,
                          const float* t_rand, const float* motion_Rs, const float* motion_Ts, const float* vol,
                          const float* bbox_min, const float* bbox_scale, const float* hann_w, const void* nr_packed,
                          const void* cnl_packed, const float* bgcolor, int mode, float cull_eps, int64_t N, int S,
                          int B, int G, int64_t chunk, void* workspace, size_t workspace_bytes, float* rgb, float* alpha,
                          float* depth, float* weights_on_rays, float* rgb_on_rays, float* cnl_xyz, float* cnl_rgb,
                          float* cnl_weight, float* xyz_on_rays, float* bmw, float* offsets, void* side_stream,
                          void* const* events, void* const* mlp_events, void* stream);

/* =============================== training (backward) ===============================
 * The reference trains through torch.autograd over the ops above (trainer.py:206-220).
 * Here: the forward runs the *_fwd_train variants (either arithmetic mode) which also save the
 * positional encodings, the post-ReLU activation matrices and their sign masks; the dX chain of
 * each MLP is one register-resident kernel (hnrf_*_bwd), the weight gradients come from
 * hnrf_mlp_dw, and the stages around the MLPs have the kernels below. */

/* pe_out [P,63] (columns in fourier.py order), acts [8][P][256] post-ReLU outputs of
 * pts_linears.{0..14}, relu_bits [8][P][8] uint32 (16-byte aligned): the sign masks of acts in the
 * register order of hnrf_canonical_bwd (opaque to the caller). */
int hnrf_canonical_fwd_train(const float* xyz, const void* packed, int mode, int64_t P,
                             float* raw, float* pe_out, float* acts, uint32_t* relu_bits, void* stream);
/* pe_out [P,36] (hannw_fourier.py order, window weights applied), acts [6][P][128],
 * relu_bits [6][P][4] uint32. */
int hnrf_nonrigid_fwd_train(const float* x_skel, const float* hann_w, const void* packed,
                            int mode, int64_t P, float* xyz, float* offsets,
                            float* pe_out, float* acts, uint32_t* relu_bits, void* stream);

/* Backward of hnrf_composite_fwd w.r.t. raw and fg_mask (autograd of network.py:355-379).
 *  g_rgb [R,3]; g_alpha, g_depth [R] or NULL.  Outputs d_raw [R,S,4], d_mask [R,S]. */
int hnrf_composite_bwd(const float* raw, const float* fg_mask, const float* z_vals,
                       const float* rays_d, const float* bgcolor,
                       const float* g_rgb, const float* g_alpha, const float* g_depth,
                       int64_t R, int S, float* d_raw, float* d_mask, void* stream);

/* Backward of the positional encodings w.r.t. the position (fourier.py / hannw_fourier.py).
 *  g [P, 3*include_input + 6*n_bands]; hann_w [n_bands] or NULL; dx [P,3] written or
 *  accumulated into. */
int hnrf_pe_bwd(const float* x, const float* g, const float* hann_w, int64_t P, int n_bands,
                int include_input, int accumulate, float* dx, void* stream);

/* Backward of hnrf_sample_warp_fwd w.r.t. the weight volume and the motion bases
 * (autograd of network.py:392-444, including grid_sample's gradient w.r.t. the grid).
 *  z_vals, x_skel, fg_mask: outputs of the forward; g_x_skel [R,S,3], g_mask [R,S].
 *  Outputs (overwritten): d_vol [B,G,G,G], d_Rs [B,3,3], d_Ts [B,3]. */
int hnrf_sample_warp_bwd(const float* rays_o, const float* rays_d, const float* z_vals,
                         const float* motion_Rs, const float* motion_Ts, const float* vol,
                         const float* bbox_min, const float* bbox_scale,
                         const float* x_skel, const float* fg_mask,
                         const float* g_x_skel, const float* g_mask,
                         int64_t R, int S, int B, int G,
                         float* d_vol, float* d_Rs, float* d_Ts, void* stream);

/* Per-frame kinematics: MotionBasisComputer.forward (core/utils/network_util.py:125-156) and its backward as one
 * single-wave kernel each (in PyTorch: 23 dependent 4x4 matmuls, a batched inverse and their autograd twins).
 *  dst_Rs [24,3,3], dst_Ts [24,3], cnl_gtfms [24,4,4] -> Rs [24,3,3], Ts [24,3] = (cnl_gtfms_i A_i^-1)[:3,:3 | :3,3],
 *  A_i = A_parent(i) [R_i | T_i] along the SMPL tree; fp64 internally.  saved (nullable): hnrf_motion_basis_saved_bytes()
 *  bytes, 8-byte aligned, for the backward.  B must be 24.
 *  bwd: g_Rs, g_Ts (gradients at the outputs) -> d_dst_Rs [24,3,3], d_dst_Ts [24,3]. */
size_t hnrf_motion_basis_saved_bytes(void);
int hnrf_motion_basis_fwd(const float* dst_Rs, const float* dst_Ts, const float* cnl_gtfms, int B, float* Rs, float* Ts,
                          void* saved, void* stream);
int hnrf_motion_basis_bwd(const float* g_Rs, const float* g_Ts, const float* dst_Rs, const float* dst_Ts,
                          const float* cnl_gtfms, int B, const void* saved, float* d_dst_Rs, float* d_dst_Ts, void* stream);
/* The same with the pose refinement in front folded in (BodyPoseRefiner's Rodrigues step, core/utils/network_util.py:57-83,
 * and the correction of core/nets/human_nerf/network.py:677-688): dst_Rs[i] <- dst_Rs[i] Rodrigues(rvec[i-1]) for the 23
 * non-root bones, theta = sqrt(1e-5 + |r|^2).  rvec [23,3] = the pose MLP's output.  bwd also returns d_rvec [23,3]
 * (d_dst_Rs is the gradient at the UNrefined rotations). */
int hnrf_refined_motion_basis_fwd(const float* rvec, const float* dst_Rs, const float* dst_Ts, const float* cnl_gtfms, int B,
                                  float* Rs, float* Ts, void* saved, void* stream);
int hnrf_refined_motion_basis_bwd(const float* g_Rs, const float* g_Ts, const float* rvec, const float* dst_Rs,
                                  const float* dst_Ts, const float* cnl_gtfms, int B, const void* saved, float* d_rvec,
                                  float* d_dst_Rs, float* d_dst_Ts, void* stream);

/* The pose refiner's MLP on one pose vector (BodyPoseRefiner.block_mlps, core/nets/human_nerf/pose_decoders/
 * mlp_delta_body_pose.py:14-41: Linear + ReLU x mlp_depth, then Linear) and its backward: one small launch per layer
 * each way (PyTorch: ~30 launches per training step for 0.5 MFLOP).
 *  W, b: HOST arrays of `layers` device pointers to the nn.Linear weights (out, in) row-major / biases; dims: HOST array
 *  of layers + 1 widths (dims[0] inputs, dims[l + 1] outputs of layer l), every width <= 256, layers <= 9.
 *  saved: hnrf_pose_mlp_saved_bytes(layers) bytes of device memory: the forward leaves the hidden activations there,
 *  the backward reads them and uses the rest as scratch.
 *  fwd: x [dims[0]] -> out [dims[layers]].
 *  bwd: g_out [dims[layers]] -> dW[l] (out, in), db[l] (HOST arrays of device pointers); d_x_parts (nullable): [8][256]
 *  floats whose first ceil(dims[1] / 32) rows sum to d_x (columns < dims[0]). */
size_t hnrf_pose_mlp_saved_bytes(int layers);
int hnrf_pose_mlp_fwd(const float* x, const float* const* W, const float* const* b, const int* dims, int layers, float* out,
                      float* saved, void* stream);
int hnrf_pose_mlp_bwd(const float* g_out, const float* x, const float* const* W, const float* const* b, const int* dims,
                      int layers, float* saved, float* const* dW, float* const* db, float* d_x_parts, void* stream);

/* Weight / bias gradient of one nn.Linear inside the two MLPs (autograd of the Linear layers of
 * canonical_mlps/mlp_rgb_sigma.py and non_rigid_motion_mlps/mlp_offset.py under trainer.py:139-170):
 *   dW[o][i] = sum_s dZ[s][o] X[s][i]  for o < n_out, i < n_in;   db[o] = sum_s dZ[s][o]  (db may be NULL).
 *  dZ [P, ldz >= n_out], X [P, ldx >= n_in] row-major fp32; dW written with row stride ldw (so the two
 *  column blocks of a skip layer's weight are two calls).  Built shapes: n_out 128 | 256 with
 *  n_in 128 | 256 (X 16-byte aligned, ldx % 4 == 0) or n_in <= 64 (any ldx: the PE matrices);
 *  n_out <= 4 (the sigma/rgb and offset heads) with n_in 128 | 256.
 *  mode HNRF_MLP_F32: fp32 MFMA.  HNRF_MLP_F16X3: split-f16 MFMA at fp32-class accuracy for the matrix-shaped
 *  layers (n_out, n_in in {128, 256}; other shapes silently use the fp32 kernels); needs dz_amax = n_amax device
 *  floats whose maximum is >= max |dZ| (one row of hnrf_*_bwd's dz_amax), and |X| <= 65504.  Deterministic (fixed-order slice reduction).
 *  workspace: hnrf_mlp_dw_workspace_bytes (covers both modes). */
size_t hnrf_mlp_dw_workspace_bytes(int64_t P, int n_out, int n_in);
int hnrf_mlp_dw(const float* dZ, int64_t ldz, const float* X, int64_t ldx, int64_t P, int n_out, int n_in,
                int mode, const float* dz_amax, int n_amax, float* dW, int64_t ldw, float* db, void* workspace,
                size_t workspace_bytes, void* stream);

/* The same weight / bias gradient from HALF-PRECISION operands (training with the activations and dZ stored as
 * f16: half the HBM traffic of the step's two largest buffers).  dZ [P, ldz] and X [P, ldx] are row-major f16
 * (16-byte aligned, strides multiples of 8 halves, X rows padded with zeros to 64 / 128 / 256 columns); dZ may carry a
 * power-of-two scale: dz_scale (nullable device scalar) is divided out of dW and db.  One f16 MFMA per product,
 * fp32 accumulation over the samples; both operands reach the matrix pipe through gfx950's transposed LDS read
 * (ds_read_b64_tr_b16).  Each operand is rounded to 11 bits, unbiased: the relative error of a sum over N samples is
 * ~2^-12 / sqrt(N).  n_out <= 4 (heads): dZ is the fp32 [P, n_out] gradient at the head output, X the f16 activations.
 * Same shapes, workspace rules and determinism as hnrf_mlp_dw.
 * layout: 0 = both matrices row-major; HNRF_DWH_DZ_BLOCKED / HNRF_DWH_X_BLOCKED = the matrix is in the BLOCKED layout
 * the HNRF_MLP_F16X3_H training kernels write (32-sample blocks of [32-feature tile][4 groups][2][32 samples][4 halves],
 * layers padded to a multiple of 128 samples; row strides are ignored for it).  Built: none, dZ only, both. */
#define HNRF_DWH_DZ_BLOCKED 1
#define HNRF_DWH_X_BLOCKED  2
size_t hnrf_mlp_dw_h_workspace_bytes(int64_t P, int n_out, int n_in);
int hnrf_mlp_dw_h(const void* dZ, int64_t ldz, const void* X, int64_t ldx, int64_t P, int n_out, int n_in, int layout,
                  const float* dz_scale, float* dW, int64_t ldw, float* db, void* workspace, size_t workspace_bytes,
                  void* stream);

/* Backward through all layers of one MLP (the dX chain of autograd over mlp_rgb_sigma.py / mlp_offset.py),
 * register-resident like the forward, with the backward of the positional encoding fused.
 *  *_bwd_pack: transposed weight image from the same nn.Linear weights hnrf_*_pack takes (re-pack after
 *  every parameter update).
 *  canonical: xyz [P,3], d_raw [P,4] (16-byte aligned), relu_bits [8][P][8] from hnrf_canonical_fwd_train ->
 *    dZ [8][P][256] (gradient at every hidden layer's pre-activation: the dZ operand of hnrf_mlp_dw) and
 *    d_xyz [P,3].
 *  non-rigid: x_skel [P,3], hann_w [6], d_xyz [P,3], relu_bits [6][P][4] from hnrf_nonrigid_fwd_train ->
 *    dZ [6][P][128] and d_x_skel [P,3] = d_xyz + J_offset^T d_xyz  (xyz = x_skel + offset, network.py:518-530). */
size_t hnrf_canonical_bwd_packed_bytes(int mode);
size_t hnrf_nonrigid_bwd_packed_bytes(int mode);
int hnrf_canonical_bwd_pack(const float* const* weights, int mode, void* packed, void* stream);
int hnrf_nonrigid_bwd_pack(const float* const* weights, int mode, void* packed, void* stream);
/* dz_amax (nullable): [L][HNRF_AMAX_SLOTS] floats; max over row l bounds |dZ_l| (the scale input of
 * hnrf_mlp_dw in HNRF_MLP_F16X3 mode). */
#define HNRF_AMAX_SLOTS 64
/* mode HNRF_MLP_F16X3: the chain on the split-f16 matrix pipe; d_raw_amax / d_xyz_amax = device scalar >= the
 * largest magnitude of the incoming gradient (sets its power-of-two scale), ignored in HNRF_MLP_F32. */
int hnrf_canonical_bwd(const float* xyz, const float* d_raw, const uint32_t* relu_bits, const void* packed,
                       int mode, const float* d_raw_amax, int64_t P, float* dZ, float* d_xyz, float* dz_amax,
                       void* stream);
int hnrf_nonrigid_bwd(const float* x_skel, const float* hann_w, const float* d_xyz, const uint32_t* relu_bits,
                      const void* packed, int mode, const float* d_xyz_amax, int64_t P, float* dZ, float* d_x_skel,
                      float* dz_amax, void* stream);

/* =============================== in front of the path ===============================
 * Ray generation + bbox intersection + order-preserving compaction (get_rays_from_KRT,
 * core/utils/camera_util.py:132-159; rays_intersect_3d_bbox, camera_util.py:162-208; called per frame at
 * core/data/human_nerf/freeview.py:220-230 and its siblings).
 *  Kinv [3,3] = inverse intrinsics, R [3,3], T [3] = extrinsics, bbox_min / bbox_max [3] (unpadded; the 1 cm pad
 *  is applied inside), all float32 device pointers.  Outputs, in pixel order: ray_mask [H*W] (1 = the ray crosses
 *  the box), and for the *count kept rays rays_o / rays_d [count,3] (direction un-normalised, components clamped
 *  to 1e-5 like the reference), near / far [count].  Size the ray buffers for H*W. */
size_t hnrf_gen_rays_workspace_bytes(int H, int W);
int hnrf_gen_rays(const float* Kinv, const float* R, const float* T, const float* bbox_min, const float* bbox_max,
                  int H, int W, float* rays_o, float* rays_d, float* near, float* far, uint8_t* ray_mask,
                  int* count, void* workspace, size_t workspace_bytes, void* stream);

/* ---- image pre-processing of Dataset.load_image (core/data/human_nerf/train.py:351-417, freeview.py:137-166) ----
 * The reference decodes a frame's image and mask PNGs and runs cv2.undistort on both (train.py:366-371: every
 * prepared ZJU-MoCap camera has `distortions`, tools/prepare_zju_mocap/prepare_dataset.py:172-176), composites
 * mask/255 * image + (1 - mask/255) * bgcolor in float64 (train.py:406) and, for resize_img_scale != 1 (0.5 in
 * every 387 / wild yaml), cv2.resize with INTER_LANCZOS4 (image) and INTER_LINEAR (mask) (train.py:408-417).
 * humannerf_amd/imageproc.py is the host statement of the same OpenCV functions (cv2 is not importable: parity with
 * OpenCV's binaries is unpinned; the two statements agree bit for bit).
 *
 * hnrf_undistort_image: src / dst uint8 [H,W,C] (C <= 4, not in place).  cam = {fx, fy, u0, v0, k1, k2, p1, p2, k3}
 *  and ir = [n_stripes][9] row-major inverses of K with its principal point moved to the stripe's first row
 *  (cv2.undistort works in stripes of stripe_rows = min(max(1, 4096 / W), H) rows), both float64 DEVICE arrays.
 *  Fixed-point map (1/32 pixel), bilinear taps with remap's 15-bit weights, constant 0 outside the image. */
int hnrf_undistort_image(const uint8_t* src, int H, int W, int C, const double* cam, const double* ir, int n_stripes,
                         int stripe_rows, uint8_t* dst, void* stream);
/* hnrf_composite_windows: orig / alpha uint8 [Hs,Ws,3] (alpha in 0..255), bgcolor float32 [3] in 0..255 ->
 *  out float32 [n_win, ph, pw, 3] = composite / 255 for the windows whose top-left DESTINATION pixels are
 *  win_xy[n_win][2] = (x0, y0) (int32, device; windows must lie inside [Hd, Wd]).  resize = 0: destination grid =
 *  source grid (Hd == Hs, Wd == Ws, tables ignored).  resize = 1: INTER_LANCZOS4 from the coefficient tables of
 *  hal::resize's setup loop -- xofs [Wd] / yofs [Hd] int32 (index of the tap left of the sample point), xw [Wd][8] /
 *  yw [Hd][8] float32 -- horizontal then vertical pass over the float64 composite, taps clamped to the border.
 *  A training item's 6 windows of 32x32 (cfg.patch) or a whole image (one window) alike. */
int hnrf_composite_windows(const uint8_t* orig, const uint8_t* alpha, int Hs, int Ws, const float* bgcolor, int resize,
                           const int* xofs, const float* xw, const int* yofs, const float* yw, int Hd, int Wd,
                           const int* win_xy, int n_win, int ph, int pw, float* out, void* stream);
/* hnrf_resize_mask: channel `channel` of the uint8 mask [Hs,Ws,3], divided by 255, through cv2.resize INTER_LINEAR ->
 *  out float32 [Hd,Wd] (the dataset only asks whether it is > 0: train.py:620-631).  mode 1: two-tap tables
 *  (xw [Wd][2], yw [Hd][2]); mode 2: the 2x2 box mean hal::resize substitutes at scale exactly 1/2. */
int hnrf_resize_mask(const uint8_t* alpha, int Hs, int Ws, int channel, int mode, const int* xofs, const float* xw,
                     const int* yofs, const float* yw, int Hd, int Wd, float* out, void* stream);

/* ---- weight-volume decoder glue (MotionWeightVolumeDecoder, mweight_vol_decoders/deconv_vol_decoder.py:25-33;
 * ConvDecoder3D, core/utils/network_util.py:12-50) ----
 * hnrf_deconv_fold: the scatter half of nn.ConvTranspose3d(kernel 4, stride 2, padding 1) on a batch-1 volume.
 *  col [D*H*W, cout*64] = x^T W (one GEMM on the weight's native (cin, cout, 4,4,4) layout, done by the caller's BLAS),
 *  bias [cout] or NULL -> out [cout, 2D, 2H, 2W]: out[co, 2d-1+kd, 2h-1+kh, 2w-1+kw] += col[(d,h,w), co*64 + kd*16+kh*4+kw],
 *  written as a gather per output voxel (deterministic, no atomics). */
int hnrf_deconv_fold(const float* col, const float* bias, int cout, int D, int H, int W, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HNRF_H */
