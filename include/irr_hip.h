/*
 * irr_hip.h -- C ABI of libirr_hip.so, the MI355X (gfx950) kernels behind the IRR-PWC hot path.
 *
 * This is the drop-in boundary.  It replaces, for visinf/irr:
 *   - the legacy pybind module ``correlation_cuda`` (models/correlation_package/correlation_cuda.cc:8-14,
 *     86-93, 165-168: forward(in1,in2,rbot1,rbot2,out, pad,k,md,s1,s2,mult) / backward(...)), and
 *   - every stock-torch operator call site of ``PWCNet.forward`` (models/IRR_PWC.py:51-184), see the
 *     per-function citations below.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer to fp32 data unless stated;
 *   - tensors are NCHW with dense H*W planes: element (b,c,y,x) of tensor T lives at
 *     T + b*T_bs + c*H*W + y*W + x, where the batch stride ``T_bs`` (in elements) is passed
 *     explicitly so that a tensor may be a channel slice of a larger buffer (the DenseNet concat
 *     buffers of FlowEstimatorDense are written in place, no torch.cat copies);
 *   - ``stream`` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises; every launcher issues
 *     KERNELS only (zero fills included: no hipMemsetAsync / hipMemcpyAsync), so it may be captured into a hipGraph -- a memset
 *     node was observed not to be ordered against the kernels around it (profiles/r4_graph_bisect.txt);
 *   - no allocation inside: the caller owns every buffer (the reference instead resize_()s and
 *     fill_(0)s inside the C++ glue, correlation_cuda.cc:34-40,104-112);
 *   - return value: 0 on success, otherwise the hipError_t of the failing call / launch
 *     (the reference returns 1 on success and raises AT_ERROR on failure, correlation_cuda.cc:78-83);
 *     IRR_EINVAL (-22) for arguments outside the supported range.
 *   - launchers are re-entrant: they may be called concurrently from several host threads (backward runs on
 *     autograd's threads) and on several streams / devices.  Host-side state is limited to (a) read-only caches of
 *     device properties and occupancy numbers, (b) thread-local hand-over variables inside one call, and (c) ONE
 *     process-wide, atomic routing policy, ``irr_conv_x3_set_min_blocks`` (a tuning knob that decides which kernel
 *     family a problem is routed to -- both families compute the same function, so a concurrent change can never
 *     produce a wrong result, only a different choice; production code never touches it);
 *   - kernels are launched on ``stream`` from the CALLING thread's current HIP device: make the device that owns the
 *     stream and the buffers current first (the host binding does -- irr_amd/hip.py:device_of -- as the reference does
 *     with torch.cuda.device_of, models/correlation_package/correlation.py:21,34).
 */
#ifndef IRR_HIP_H
#define IRR_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define IRR_EINVAL (-22)

/* library / device probe.  Returns the ABI version (increases when a signature changes). */
int irr_abi_version(void);

/* ---- 81-channel cost volume -------------------------------------------------------------------
 * out[b,(dy+4)*9+(dx+4),y,x] = (1/C) * sum_c f1[b,c,y,x] * f2[b,c,y+dy,x+dx]   (zero outside), dy,dx in [-4,4]
 * == compute_cost_volume (models/pwc_modules.py:42-62, call sites models/IRR_PWC.py:90-91)
 * == Correlation(pad_size=4,kernel_size=1,max_displacement=4,stride1=1,stride2=1)
 *    (models/correlation_package/correlation.py:47-61, correlation_cuda_kernel.cu:41-114).
 * fuse_lrelu != 0 additionally applies LeakyReLU(0.1) (models/IRR_PWC.py:94-95).
 */
int irr_corr81_fwd_f32(const float* f1, const float* f2, float* out,
                       int B, int C, int H, int W,
                       long f1_bs, long f2_bs, long out_bs,
                       int fuse_lrelu, void* stream);

/* Gradients of the above (correlation_cuda_kernel.cu:116-300, both in gather form).
 * ``out`` (nullable) is the forward result when fuse_lrelu was set: gout is then multiplied by the
 * LeakyReLU derivative (1 where out>0, else 0.1).  g1/g2 are fully overwritten. */
int irr_corr81_bwd_f32(const float* f1, const float* f2, const float* gout, const float* out,
                       float* g1, float* g2,
                       int B, int C, int H, int W,
                       long f1_bs, long f2_bs, long gout_bs, long out_bs, long g1_bs, long g2_bs,
                       void* stream);

/* The legacy operator at ANY parameter point (models/correlation_package/correlation.py:47-61; forward arithmetic
 * correlation_cuda_kernel.cu:41-114, output shape correlation_cuda.cc:23-32): P = input zero-padded by pad, kr = (k - 1) / 2 (k odd),
 * dr = md / s2, D = 2 dr + 1, (y1, x1) = (oy s1 + md, ox s1 + md):
 *   out[n, (tj+dr) D + (ti+dr), oy, ox] = 1/(k k C) sum_{j,i in [-kr,kr]} sum_c P1[n,c,y1+j,x1+i] P2[n,c,y1+tj s2+j,x1+ti s2+i]
 * with OH = ceil((H + 2 pad - 2 (kr + md)) / s1) (irr_corr_general_out_shape).  Backward = the exact adjoint (for k = 1, s1 = 1 identical
 * to correlation_cuda_kernel.cu:116-300).  General-purpose kernels (one thread per element); the IRR-PWC point (4, 1, 4, 1, 1) is
 * irr_corr81_*.  g1 / g2 nullable.  (ABI 8) */
int irr_corr_general_out_shape(int H, int W, int pad, int k, int md, int s1, int s2, int* channels, int* OH, int* OW);
int irr_corr_general_fwd_f32(const float* f1, const float* f2, float* out, int B, int C, int H, int W,
                             int pad, int k, int md, int s1, int s2, long f1_bs, long f2_bs, long out_bs, void* stream);
int irr_corr_general_bwd_f32(const float* f1, const float* f2, const float* gout, float* g1, float* g2,
                             int B, int C, int H, int W, int pad, int k, int md, int s1, int s2,
                             long f1_bs, long f2_bs, long gout_bs, long g1_bs, long g2_bs, void* stream);

/* ---- flow warping with validity mask -----------------------------------------------------------
 * WarpingLayer.forward (models/pwc_modules.py:115-133) incl. get_grid (:107-112):
 *   grid = linspace(-1,1) + flow*2/max(size_im-1,1)/div_flow ; bilinear, zeros padding,
 *   align_corners=True ; out = sample(x) * (sample(ones) >= mask_thr).
 * swap_halves != 0 (B even): sample b warps x[(b + B/2) % B] -- both flow directions run as one batch [x1; x2] whose "other
 * image" is [x2; x1]; the kernel reads (and, in backward, scatters into) the other batch half instead of a swapped copy.
 * gridx[W], gridy[H] are the two torch.linspace(-1,1,n) vectors (device), so that the base grid is
 * bit-identical to the reference's.  mask_thr = 1.0 is the reference as-is.
 */
int irr_warp_fwd_f32(const float* x, const float* flow, const float* gridx, const float* gridy,
                     float* out, int B, int C, int H, int W,
                     long x_bs, long flow_bs, long out_bs,
                     int height_im, int width_im, float div_flow, float mask_thr, int swap_halves, void* stream);

/* gx (nullable) receives the scatter-add gradient w.r.t. x (zeroed inside), gflow (nullable) the
 * gradient w.r.t. flow.  No gradient flows through the mask (piecewise constant). */
int irr_warp_bwd_f32(const float* x, const float* flow, const float* gridx, const float* gridy,
                     const float* gout, float* gx, float* gflow,
                     int B, int C, int H, int W,
                     long x_bs, long flow_bs, long gout_bs, long gx_bs, long gflow_bs,
                     int height_im, int width_im, float div_flow, float mask_thr, int swap_halves, void* stream);

/* The same gradients with the gradient w.r.t. x computed OWNER-COMPUTES (round 4): every pixel of gx is gathered by the thread
 * that owns it from the output pixels whose bilinear targets include it (4-tap gather like the forward pass: no atomics, no zero
 * fill), and the gradient w.r.t. the flow by a kernel with lanes along x.  Valid while every bilinear target stays within 16 pixels
 * (per axis) of its output pixel and no pixel of gx has more than 24 contributors; a SAMPLE that violates this is detected on the device and takes the atomic scatter of
 * irr_warp_bwd_f32 inside the same call.  ws: caller-owned scratch of irr_warp_bwd_ws_elems(B, H, W) ints (any contents; only
 * needed when gx != NULL). */
long irr_warp_bwd_ws_elems(int B, int H, int W);
int irr_warp_bwd_gather_f32(const float* x, const float* flow, const float* gridx, const float* gridy,
                            const float* gout, float* gx, float* gflow,
                            int B, int C, int H, int W,
                            long x_bs, long flow_bs, long gout_bs, long gx_bs, long gflow_bs,
                            int height_im, int width_im, float div_flow, float mask_thr, int swap_halves,
                            int* ws, long ws_elems, void* stream);

/* ---- bilinear resize, align_corners=True --------------------------------------------------------
 * upsample2d_as (models/pwc_modules.py:65-67).  out = alpha * resize(x).
 */
int irr_resize_bilinear_ac_fwd_f32(const float* x, float* out, int B, int C, int H, int W, int OH, int OW,
                                   long x_bs, long out_bs, float alpha, void* stream);
/* gx = alpha * resize^T(gout); gx fully overwritten (gather form, deterministic). */
int irr_resize_bilinear_ac_bwd_f32(const float* gout, float* gx, int B, int C, int H, int W, int OH, int OW,
                                   long gout_bs, long gx_bs, float alpha, void* stream);
/* (ABI 12) accumulate = 1: gx += alpha * resize^T(gout) -- the gradients of one tensor that was resized to several sizes meet in one
 * buffer (calls on one stream are ordered; no atomics); 0: irr_resize_bilinear_ac_bwd_f32. */
int irr_resize_bilinear_ac_bwd_acc_f32(const float* gout, float* gx, int B, int C, int H, int W, int OH, int OW,
                                       long gout_bs, long gx_bs, float alpha, int accumulate, void* stream);
/* The same with align_corners=False (half-pixel centres, ATen's F.interpolate(x, [OH, OW], mode="bilinear")): the fallback
 * of upsample_factor2 when the nearest-x2 map does not have the guide's size, i.e. odd pyramid sizes
 * (models/irr_modules.py:21-27; Sintel 436x1024 -> 218, 109, 55, ...). */
int irr_resize_bilinear_hp_fwd_f32(const float* x, float* out, int B, int C, int H, int W, int OH, int OW,
                                   long x_bs, long out_bs, float alpha, void* stream);
int irr_resize_bilinear_hp_bwd_f32(const float* gout, float* gx, int B, int C, int H, int W, int OH, int OW,
                                   long gout_bs, long gx_bs, float alpha, void* stream);

/* ---- convolution family (fp32 MFMA implicit GEMM) -----------------------------------------------
 * conv() helper (models/pwc_modules.py:8-19, models/irr_modules.py:7-18): Conv2d(k in {1,3}, stride in {1,2},
 * dilation d, padding (k-1)*d/2, bias) followed by an optional LeakyReLU(0.1).
 *
 * Weights are consumed in a packed layout produced by irr_conv_pack_weights_f32:
 *   wp[((cp*KK + tap)*2 + half)*CoP + co] = w[co][2*cp+half][tap]   (0 where ci or co is padding),
 *   CoP = roundup(Cout,32), cp in [0, ceil(Cin/2)), KK = k*k.
 * With transpose != 0 the roles of Cin/Cout are swapped and the taps flipped, which turns the same
 * forward kernel into the data-gradient (stride 1 only).
 */
long irr_conv_packed_weight_elems(int Cin, int Cout, int k);
int irr_conv_pack_weights_f32(const float* w, float* wp, int Cin, int Cout, int k, int transpose, void* stream);

/* Combined data-gradient weights: rows [row_offset, row_offset+w_cout) of a packed (transposed, tap-flipped) matrix
 * with CoP columns are filled from layer weights w (w_cout, w_cin, k, k) restricted to input channels
 * [chan0, chan0+nchan).  Several layers that read the same channel range of a DenseNet buffer can thus be
 * back-propagated into that range by ONE launch of irr_conv2d_fwd_f32 over their concatenated output gradients.
 * The destination must hold irr_conv_packed_weight_elems(total_rows, nchan, k) floats, zero-initialised. */
int irr_conv_pack_weights_sub_f32(const float* w, float* wp, int w_cin, int w_cout, int k, int chan0, int nchan,
                                  int CoP, int row_offset, void* stream);

/* y = epilogue(conv(x, wp) + bias):
 *   v = acc + bias[co] (bias nullable);  if (lrelu) v = v>0 ? v : 0.1 v;
 *   y = res ? res + alpha*v : alpha*v      (res nullable; OccUpsampleNetwork residual adds,
 *                                           models/irr_modules.py:51-54, and flow + flow_res, models/IRR_PWC.py:110-114)
 *   accumulate != 0: y += previous y        (data-gradient accumulation into DenseNet gradient buffers).
 *   mask (nullable): finally y *= LeakyReLU'(mask[b,co,p]) for co < nmask -- lets a data-gradient launch hand the
 *   NEXT layer its pre-activation gradient directly (mask = that layer's saved activation), so no separate
 *   LeakyReLU-backward pass over the tensor is needed.
 * Cin >= 2.  The batch is split internally so that all in-kernel byte offsets stay below 4 GiB.
 */
int irr_conv2d_fwd_f32(const float* x, const float* wp, const float* bias, const float* res, float* y,
                       int B, int Cin, int H, int W, int Cout, int OH, int OW,
                       int k, int stride, int dil,
                       long x_bs, long y_bs, long res_bs,
                       int lrelu, float alpha, int accumulate,
                       const float* mask, long mask_bs, int nmask, void* stream);

/* Which template instantiation irr_conv2d_fwd_f32 launches for this problem, as MT*100 + NT*10 + k
 * (conv_fwd_kernel<MT,NT,k>); used to label bench.py's roofline line and to find the kernel in rocprof output. */
int irr_conv2d_fwd_variant(int B, int Cout, int OH, int OW, int k);

/* ---- fp32-faithful 3x3 stride-1 convolution on the bf16 matrix pipe (csrc/conv_x3.hip) --------------------
 * Same operator and epilogue as irr_conv2d_fwd_f32 for k = 3, stride = 1, Cin >= 16.  Every fp32 operand is split
 * exactly into three bf16 pieces and each product is accumulated in fp32 from its six leading piece products
 * (error <= 2^-24 |a*b| per product: the error class of an fp32 FMA chain), which runs at 6/16 of the cost of the
 * fp32 MFMA.  Weights are consumed pre-split:
 *   wq[(((chunk*9 + tap)*3 + piece)*CoT + cot)*64 + lane] = 8 bf16 (16 B): output channel cot*32 + (lane & 31),
 *   input channels chunk*16 + 8*(lane >> 5) + 0..7  (the last chunk covers [Cin-16, Cin) when Cin % 16 != 0, with
 *   zeros for channels an earlier chunk already covered);  CoT = ceil(Cout/32).
 * transpose != 0: w is the ORIGINAL (Cin, Cout, 3, 3) tensor used transposed + tap-flipped (stride-1 data gradient).
 * irr_conv_pack_weights_x3_sub: the combined-matrix variant of irr_conv_pack_weights_sub_f32 (rows of other layers
 * are left untouched; the destination must be zero-initialised; total_rows % 16 == 0, row_offset % 8 == 0).
 * irr_conv2d_x3_eligible: non-zero (= a code naming the template instantiation) when irr_conv2d_fwd_x3 accepts the
 * problem AND it is large enough to fill the chip; 0 -> use irr_conv2d_fwd_f32. */
long irr_conv_x3_packed_bytes(int Cin, int Cout);
int irr_conv_pack_weights_x3(const float* w, void* wq, int Cin, int Cout, int transpose, void* stream);
int irr_conv_pack_weights_x3_sub(const float* w, void* wq, int w_cin, int w_cout, int total_rows, int chan0,
                                 int nchan, int row_offset, void* stream);
int irr_conv2d_x3_eligible(int B, int Cin, int H, int W, int Cout, int k, int stride, int dil);

/* ---- batched weight packing ------------------------------------------------------------------------------------------------
 * After an optimizer step every packed copy of every conv weight is stale; instead of one launch per copy (~250 per step) the host
 * keeps a table of pack jobs on the device and refreshes all of them in ONE dispatch.  A job record (irr_conv_pack_job_bytes()
 * bytes, layout private to the library) is produced in HOST memory by the builder that mirrors the single-job launcher of the
 * same name; it returns the number of 256-thread blocks the job needs (negative: IRR_EINVAL) and leaves the record's first-block
 * field (a long at byte offset irr_conv_pack_job_block0_offset()) zero: the caller lays the jobs out back to back (exclusive prefix sum of the block counts),
 * copies the table to the device and calls irr_conv_pack_batch(table, njobs, total blocks, stream). */
int irr_conv_pack_job_bytes(void);
int irr_conv_pack_job_block0_offset(void);
long irr_conv_pack_job_f32(void* job, const float* w, float* wp, int Cin, int Cout, int k, int transpose);
long irr_conv_pack_job_sub_f32(void* job, const float* w, float* wp, int w_cin, int w_cout, int k, int chan0, int nchan,
                               int CoP, int row_offset);
long irr_conv_pack_job_x3(void* job, const float* w, void* wq, int Cin, int Cout, int transpose);
long irr_conv_pack_job_x3_sub(void* job, const float* w, void* wq, int w_cin, int w_cout, int total_rows, int chan0,
                              int nchan, int row_offset);
int irr_conv_pack_batch(const void* jobs, int njobs, long nblocks, void* stream);
/* tuning knob (tests use 0 to exercise the kernel on small problems): minimum number of blocks a launch must have
 * for irr_conv2d_x3_eligible to accept it; n < 0 only queries.  Returns the previous value (default 384). */
int irr_conv_x3_set_min_blocks(int n);
int irr_conv2d_fwd_x3(const float* x, const void* wq, const float* bias, const float* res, float* y,
                      int B, int Cin, int H, int W, int Cout, int dil,
                      long x_bs, long y_bs, long res_bs,
                      int lrelu, float alpha, int accumulate,
                      const float* mask, long mask_bs, int nmask, void* stream);
/* irr_conv2d_fwd_x3 with a SECOND output, for the problems the streaming 32-channel kernel takes (irr_conv2d_x3_eligible ==
 * 9001; anything else, or res == NULL, is rejected with IRR_EINVAL): y2 = alpha * act(conv(x) + bias) and y = res + y2.
 * A skip connection whose branch output is needed again in backward -- x_init + res_end_conv(x_res),
 * models/irr_modules.py:54 -- then costs no elementwise pass over two 32-channel full-resolution maps. */
int irr_conv2d_fwd_x3_dual(const float* x, const void* wq, const float* bias, const float* res, float* y, float* y2,
                           int B, int Cin, int H, int W, int Cout, int dil,
                           long x_bs, long y_bs, long res_bs, long y2_bs, int lrelu, float alpha, void* stream);
/* Small pyramid levels (launches of fewer blocks than the chip has slots): the same kernel with blockIdx.z splitting the
 * 16-channel chunks (K); the slices store raw partial sums into ws and a second kernel sums them and applies the epilogue.
 * irr_conv2d_fwd_x3_ws_elems: floats of scratch the problem needs (0 = runs unsplit, use irr_conv2d_fwd_x3).
 * irr_conv2d_fwd_x3 itself never splits (without scratch the problem runs unsplit: same results class, slower). */
long irr_conv2d_fwd_x3_ws_elems(int B, int Cin, int H, int W, int Cout, int dil);
int irr_conv2d_fwd_x3_splitk(const float* x, const void* wq, const float* bias, const float* res, float* y,
                             int B, int Cin, int H, int W, int Cout, int dil,
                             long x_bs, long y_bs, long res_bs,
                             int lrelu, float alpha, int accumulate,
                             const float* mask, long mask_bs, int nmask, float* ws, long ws_elems, void* stream);

/* dW[co][ci][tap] += sum_{b,y,x} gy[b,co,y,x] * x[b,ci,y*stride+(ty-pad), x*stride+(tx-pad)]
 * gw is the plain (Cout,Cin,k,k) tensor and is ACCUMULATED into (caller zeroes it when it wants "=").
 * ws: caller-owned scratch of ws_elems >= irr_conv2d_wgrad_ws_elems(...) floats (same shape arguments; a smaller scratch is
 * rejected with IRR_EINVAL before anything is launched): every block column of the
 * split-K launch stores its partial [co][tap][ci] image there and a second kernel adds them to gw in a fixed order
 * (no atomics, no zero-fill; the weight gradient is bit-reproducible).
 * gbias (nullable): gbias[co] += sum_{b,y,x} gy[b,co,y,x] (the bias gradient, taken from the staged gy tiles).
 * alpha scales both results (residual branches y = x + alpha*conv(t): models/irr_modules.py:51-53). */
long irr_conv2d_wgrad_ws_elems(int B, int Cin, int H, int W, int Cout, int OH, int OW, int k, int stride, int dil,
                               long x_bs, long gy_bs);
int irr_conv2d_wgrad_f32(const float* x, const float* gy, float* gw, float* ws, float* gbias, float alpha,
                         int B, int Cin, int H, int W, int Cout, int OH, int OW,
                         int k, int stride, int dil, long x_bs, long gy_bs, long ws_elems, void* stream);

/* Weight gradient on the bf16 matrix pipe with the exact 3-way split of conv_x3 (csrc/conv_wgrad_x3.hip): same contract
 * as irr_conv2d_wgrad_f32 for k = 3, stride = 1, dilation = 1, W % 4 == 0 (gw accumulated, gbias nullable, alpha
 * scales both) -- except for the scratch: ws must hold irr_conv2d_wgrad_x3_ws_elems(Cin, Cout) floats (one partial
 * [Cout][9][Cin] image per block column; the partials are summed in a fixed order, so the weight gradient is
 * bit-reproducible and needs neither atomics nor a zeroed workspace).  irr_conv2d_wgrad_x3_eligible: non-zero when the
 * problem is accepted and large enough to pay off; 0 -> use irr_conv2d_wgrad_f32. */
int irr_conv2d_wgrad_x3_eligible(int B, int Cin, int H, int W, int Cout, int k, int stride, int dil);
long irr_conv2d_wgrad_x3_ws_elems(int Cin, int Cout);
int irr_conv2d_wgrad_x3(const float* x, const float* gy, float* gw, float* ws, float* gbias, float alpha,
                        int B, int Cin, int H, int W, int Cout, long x_bs, long gy_bs, void* stream);
/* ---- deferred fold of the partial images ---------------------------------------------------------------------------------------
 * Every irr_conv2d_wgrad_{f32,x3,x3_dil} launch ends with a small kernel that folds the partial images of its scratch into gw.
 * Between irr_wgrad_defer_begin(jobs, capacity) and irr_wgrad_defer_end() ON THE CALLING THREAD those launchers append a job
 * record (irr_wgrad_job_bytes() bytes, layout private to the library) to the HOST array `jobs` instead (a launch whose scratch is
 * reused by a second batch slice, or one that finds the array full, still folds immediately); irr_wgrad_defer_end returns the
 * number of records appended.  The caller keeps ws and gw alive and runs irr_wgrad_reduce_batch(jobs, njobs <=
 * irr_wgrad_reduce_batch_max(), stream) on the same stream before gw is read; two jobs of one batch must not share a gw
 * (IRR_EINVAL).  Same arithmetic and summation order as the immediate fold: results are bit-identical. */
int irr_wgrad_job_bytes(void);
int irr_wgrad_reduce_batch_max(void);
int irr_wgrad_defer_begin(void* jobs, int capacity);
int irr_wgrad_defer_end(void);
int irr_wgrad_reduce_batch(const void* jobs, int njobs, void* stream);
/* the dilated layers of the context networks (dil in {2,4,8,16}, accepted when irr_conv2d_wgrad_x3_eligible says so) */
int irr_conv2d_wgrad_x3_dil(const float* x, const float* gy, float* gw, float* ws, float* gbias, float alpha,
                            int B, int Cin, int H, int W, int Cout, int dil, long x_bs, long gy_bs, void* stream);

/* ---- "h2": the conv_x3 / conv_wgrad_x3 kernels with operands as TWO fp16 pieces of scaled values (round 4) ------------------
 * x * 2^e = hi + lo (fp16 each, round-to-nearest; 23 significant bits, absolute error <= 2^-25 in scaled units for the small
 * values of a tensor), a * b ~= ah*bh + ah*bl + al*bh accumulated in fp32 by v_mfma_f32_32x32x16_f16: half the matrix work of
 * the bf16x3 form, the same error class (dropped terms <= 2^-24 |ab|).  fp16 has a 5-bit exponent, so every operand tensor is
 * scaled by a power of two derived ON THE DEVICE from max |.| of the whole tensor (|x| * 2^e < 2^15); the launch undoes both
 * scales before its epilogue.  The maxima live in device "amax slots" (plain floats):
 *   irr_amax_f32: slot = max(slot, max |x|) over B plane-dense samples of n floats (batch stride bs); slot starts at 0.
 *   x_amax / n_amax arguments: the operand's magnitude = max over n consecutive slots (an operand assembled from several
 *   producers -- a DenseNet buffer -- carries one slot per part).  y_amax (nullable): the launch folds max |y| of what it
 *   stores into that slot (atomic max), so a chain of launches needs no separate pass over its activations.
 * Weights: irr_conv_pack_weights_h2 / _h2_sub = the x3 packers with two pieces per fragment; amax = device scalar >= max |w|
 * over every weight that goes into wq (one scale per packed matrix; its exponent is stored in the 16-B unit behind the last
 * fragment: irr_conv_h2_packed_bytes includes it).  irr_conv2d_h2_eligible: the code of irr_conv2d_x3_eligible (9001 = the
 * streaming 32-channel kernel, which has the same two forms).  irr_conv2d_fwd_h2: contract of irr_conv2d_fwd_x3_splitk (ws
 * nullable); irr_conv2d_fwd_h2_dual: contract of irr_conv2d_fwd_x3_dual (y_amax bounds y, the sum).
 * irr_conv2d_wgrad_h2: contract of irr_conv2d_wgrad_x3 (dil == 1) / irr_conv2d_wgrad_x3_dil (dil > 1), same scratch and fold.
 * Dynamic range (ABI 7): the activation-side operand of irr_conv2d_fwd_h2 / _dual (x, or gy of a data gradient) is carried as
 * hi + 2^-11 lo' with the low piece scaled UP by 2^11 -- 22-23 significant bits for every ELEMENT within 2^29 (5e8) of the
 * tensor's maximum, absolute 2^-36 of the scaled range below; the weight side multiplies its high piece by 2^-11 in registers
 * (exact within 2^18 of the matrix maximum).  The weight gradient has two activation operands: the one in the kernel's x role
 * gets the scaled-up low piece (2^28 : 1), the other the plain pair (2^17 : 1 at full precision, absolute 2^-25 of its scale
 * below -- per CHANNEL with irr_conv2d_wgrad_h2_ch, ABI 11); irr_conv2d_wgrad_h2_robust_side says which is which for a problem
 * (1: x, 0: gy). */
int irr_amax_f32(const float* x, int B, long n, long bs, float* slot, void* stream);
long irr_conv_h2_packed_bytes(int Cin, int Cout);
int irr_conv_pack_weights_h2(const float* w, void* wq, int Cin, int Cout, int transpose, const float* amax, void* stream);
int irr_conv_pack_weights_h2_sub(const float* w, void* wq, int w_cin, int w_cout, int total_rows, int chan0,
                                 int nchan, int row_offset, const float* amax, void* stream);
long irr_conv_pack_job_h2(void* job, const float* w, void* wq, int Cin, int Cout, int transpose, const float* amax);
long irr_conv_pack_job_h2_sub(void* job, const float* w, void* wq, int w_cin, int w_cout, int total_rows, int chan0,
                              int nchan, int row_offset, const float* amax);
int irr_conv2d_h2_eligible(int B, int Cin, int H, int W, int Cout, int k, int stride, int dil);
int irr_conv2d_fwd_h2(const float* x, const void* wq, const float* bias, const float* res, float* y,
                      int B, int Cin, int H, int W, int Cout, int dil,
                      long x_bs, long y_bs, long res_bs,
                      int lrelu, float alpha, int accumulate,
                      const float* mask, long mask_bs, int nmask, float* ws, long ws_elems,
                      const float* x_amax, int n_amax, float* y_amax, void* stream);
/* (ABI 12, end of round 6) irr_conv2d_fwd_h2 whose K-split launches (problems with irr_conv2d_fwd_x3_ws_elems > 0: the small pyramid levels)
 * finish INSIDE the launch: kcnt = kcnt_elems >= irr_conv2d_fwd_x3_kcounters(...) ZEROED 32-bit device counters (zero again afterwards).  The
 * block that arrives last at a pixel tile sums the slices' partial images in slice order and runs the epilogue: no finishing launch, results
 * bit-identical.  kcnt == NULL (or too few counters): exactly irr_conv2d_fwd_h2. */
long irr_conv2d_fwd_x3_kcounters(int B, int Cin, int H, int W, int Cout, int dil);
int irr_conv2d_fwd_h2_kfused(const float* x, const void* wq, const float* bias, const float* res, float* y, int B,
                             int Cin, int H, int W, int Cout, int dil, long x_bs, long y_bs, long res_bs, int lrelu,
                             float alpha, int accumulate, const float* mask, long mask_bs, int nmask, float* ws,
                             long ws_elems, void* kcnt, long kcnt_elems, const float* x_amax, int n_amax, float* y_amax,
                             void* stream);
int irr_conv2d_fwd_h2_dual(const float* x, const void* wq, const float* bias, const float* res, float* y, float* y2,
                           int B, int Cin, int H, int W, int Cout, int dil,
                           long x_bs, long y_bs, long res_bs, long y2_bs, int lrelu, float alpha,
                           const float* x_amax, int n_amax, float* y_amax, void* stream);
/* ABI 9 -- LeakyReLU' masks of the streaming 32-channel kernel as BITS.  For problems with irr_conv2d_h2_eligible == 9001 and
 * Cout <= 32 (the OccUpsampleNetwork layers at 1/2 and full resolution, models/irr_modules.py:30-56; anything else: IRR_EINVAL):
 * irr_conv2d_fwd_h2 where
 *   bits_out  (forward of a conv + LeakyReLU; res / accumulate must be absent): additionally receives one bit per output element,
 *             (y > 0), irr_conv2d_x3s_mask_words(B, H, W) 32-bit words in the kernel's own tile order (opaque to the caller);
 *   mask_bits (data gradient): gx[:, :nmask] *= LeakyReLU'(.) taken from the bits a forward launch of the SAME (B, H, W) wrote,
 *             instead of re-reading the fp32 activation (mask of irr_conv2d_fwd_h2): one dword per thread and tile instead of
 *             eight 16-byte loads -- the backward of these layers is bound by HBM bytes.
 * Exactly one of the two is non-NULL. */
long irr_conv2d_x3s_mask_words(int B, int H, int W);
int irr_conv2d_fwd_h2_bits(const float* x, const void* wq, const float* bias, const float* res, float* y,
                           int B, int Cin, int H, int W, int Cout, int dil,
                           long x_bs, long y_bs, long res_bs,
                           int lrelu, float alpha, int accumulate,
                           const void* mask_bits, int nmask, void* bits_out,
                           const float* x_amax, int n_amax, float* y_amax, void* stream);
/* ABI 10 -- Winograd F(2x2, 3x3) on the fp16x2 arithmetic (csrc/conv_wino.hip; round 6, a gated experiment: tools/wino_check.py).
 * The 3x3 / stride-1 / dilation-1 conv() block (models/pwc_modules.py:8-19) with 16 instead of 36 products per 2x2 output tile:
 * U = G g G^T is formed at pack time (irr_conv_pack_weights_wino_h2: transpose = 0 forward, 1 = the stride-1 data gradient's
 * transposed + flipped matrix; amax = device scalar >= max |w|, one scale per packed matrix; irr_conv_wino_packed_bytes bytes),
 * V = B^T d B inside the launch.  irr_conv2d_wino_fwd_h2: y = alpha * act(conv(x, w) + bias), operands as in irr_conv2d_fwd_h2
 * (x_amax / n_amax slots bound |x|; y_amax nullable: receives max |y|).  irr_conv2d_wino_eligible: 1 when the launcher accepts
 * the problem. */
long irr_conv_wino_packed_bytes(int Cin, int Cout);
int irr_conv_pack_weights_wino_h2(const float* w, void* uq, int Cin, int Cout, int transpose, const float* amax, void* stream);
int irr_conv2d_wino_eligible(int B, int Cin, int H, int W, int Cout);
int irr_conv2d_wino_fwd_h2(const float* x, const void* uq, const float* bias, float* y,
                           int B, int Cin, int H, int W, int Cout, long x_bs, long y_bs,
                           int lrelu, float alpha, const float* x_amax, int n_amax, float* y_amax, void* stream);
int irr_conv2d_wgrad_h2_robust_side(int B, int Cin, int H, int W, int Cout, int dil);
int irr_conv2d_wgrad_h2(const float* x, const float* gy, float* gw, float* ws, float* gbias, float alpha,
                        int B, int Cin, int H, int W, int Cout, int dil, long x_bs, long gy_bs,
                        const float* x_amax, int nx, const float* gy_amax, int ng, void* stream);
/* ABI 11 -- one operand scale per CHANNEL for the weight gradient's gy-role operand.  The fp32 convolution of the reference
 * (models/pwc_modules.py:8-19) resolves a gradient row whatever its size; the plain fp16 pair of the operand in the kernel's gy role
 * does so only within 2^17 of the tensor's scale.  irr_amax_channels_f32: out[c] = max |x[:, c]| over B plane-dense samples of C planes
 * of hw floats (batch stride bs; accumulate = 1: max(out[c], ...) -- out holds zeros or earlier folds -- 0: out is overwritten); irr_conv2d_wgrad_h2_ch: irr_conv2d_wgrad_h2 with x_chmax (Cin floats) / gy_chmax (Cout floats) --
 * only the one of the operand that irr_conv2d_wgrad_h2_robust_side does NOT name is read (robust side 1 -> gy_chmax, 0 -> x_chmax),
 * the other may be NULL: that operand is scaled channel by channel and the scales are undone per row of dW. */
int irr_amax_channels_f32(const float* x, int B, int C, long hw, long bs, float* out, int accumulate, void* stream);
/* one-shot: the NEXT irr_conv2d_fwd_h2 / _h2_bits / _h2_dual launch of the calling thread (either kernel family; a bf16x3 launch answers
 * IRR_EINVAL) folds max |y[:, co]| of what it stores into chmax[co] (Cout zero-initialised floats) -- the same maxima without a pass over y */
int irr_conv_x3_next_chmax(float* chmax);
int irr_conv2d_wgrad_h2_ch(const float* x, const float* gy, float* gw, float* ws, float* gbias, float alpha,
                           int B, int Cin, int H, int W, int Cout, int dil, long x_bs, long gy_bs,
                           const float* x_amax, int nx, const float* gy_amax, int ng,
                           const float* x_chmax, const float* gy_chmax, void* stream);

/* ---- tiny-Cout heads (Cout <= 4, stride 1): direct VALU kernels, same contracts as the MFMA entry points -------
 * conv_last 563->2 / 562->1, context tails 32->2 / 32->1, OccUpsampleNetwork.out_convs 32->1
 * (models/pwc_modules.py:161,198,221,239; models/irr_modules.py:44).  w is the plain (Cout,Cin,k,k) tensor. */
int irr_conv2d_smallco_fwd_f32(const float* x, const float* w, const float* bias, const float* res, float* y,
                               int B, int Cin, int H, int W, int Cout, int k, int dil,
                               long x_bs, long y_bs, long res_bs, int lrelu, float alpha, int accumulate, void* stream);
int irr_conv2d_smallco_wgrad_f32(const float* x, const float* gy, float* gw, float* ws, float* gbias, float alpha,
                                 int B, int Cin, int H, int W, int Cout, int k, int dil,
                                 long x_bs, long gy_bs, void* stream);
/* weight gradient of a 3x3 layer with THREE input channels (the first pyramid conv, any stride / dilation, padding = dil):
 * same contract as irr_conv2d_wgrad_f32 (gw accumulated, ws = Cout*Cin*9 floats of scratch, gbias nullable). */
int irr_conv2d_smallci_wgrad_f32(const float* x, const float* gy, float* gw, float* ws, float* gbias, float alpha,
                                 int B, int Cin, int H, int W, int Cout, int OH, int OW, int stride, int dil,
                                 long x_bs, long gy_bs, void* stream);
/* Stride-1 data gradient of the Cout <= 2 heads (3x3, dilation d): gx[b,ci] (+)= conv_transpose(gy, w)[b,ci], then
 * gx[:, :nmask] *= LeakyReLU'(mask) (mask nullable).  w is the plain (Cout, Cin, 3, 3) tensor.  HBM-bound VALU kernel;
 * replaces the MFMA launch with K = 9*Cout of irr_conv2d_fwd_f32 (transposed pack) for these layers. */
int irr_conv2d_smallco_dgrad_f32(const float* gy, const float* w, float* gx, const float* mask,
                                 int B, int Cin, int H, int W, int Cout, int dil,
                                 long gy_bs, long gx_bs, long mask_bs, int nmask, int accumulate,
                                 float* amax, int amax_channels, void* stream);
/* (ABI 7) amax (nullable): *amax = max(*amax, max |gx[:, :amax_channels]| as stored) -- the amax slot of the fp16x2 launches that
 * consume those channels (the DenseNet backward's first gradient slice), folded in the same pass. */
/* Both forms of the same data gradient from one pass (Cout = 1, dilation 1, W % 4 == 0, batch strides multiples of 4; IRR_EINVAL
 * otherwise): gx_raw = conv_transpose(gy, w) and gx = gx_raw * LeakyReLU'(mask) over all Cin channels.  OccUpsampleNetwork's backward
 * (models/irr_modules.py:54-55: x = x_init + res_end_conv(x); out_convs(x)) needs the gradient of that sum raw (skip) and masked. */
int irr_conv2d_smallco_dgrad_dual_f32(const float* gy, const float* w, float* gx, float* gx_raw, const float* mask,
                                      int B, int Cin, int H, int W, int Cout,
                                      long gy_bs, long gx_bs, long raw_bs, long mask_bs, float* amax, void* stream);
/* (ABI 7) amax (nullable): *amax = max(*amax, max |gx|) (the masked form). */
int irr_conv2d_smallco_dgrad_dual_ch_f32(const float* gy, const float* w, float* gx, float* gx_raw, const float* mask,
                                         int B, int Cin, int H, int W, int Cout,
                                         long gy_bs, long gx_bs, long raw_bs, long mask_bs, float* amax, float* chmax, void* stream);
/* (ABI 12) the same with chmax (nullable, Cin zero-initialised floats): chmax[ci] = max(chmax[ci], max |gx[:, ci]|), the channel maxima of
 * the masked form -- the scales of the weight gradient that takes gx as its gy (irr_conv2d_wgrad_h2_ch) without a pass over it. */


/* gpre = gy * (y>0 ? 1 : 0.1) (if lrelu) ; gbias[co] += sum gpre (gbias nullable, accumulated).
 * gpre may alias gy.  amax (nullable, ABI 7): *amax = max(*amax, max |gpre|) -- the amax slot of the fp16x2 launches that read gpre. */
int irr_lrelu_bwd_bias_f32(const float* gy, const float* y, float* gpre, float* gbias,
                           int B, int C, int HW, long gy_bs, long y_bs, long gpre_bs,
                           int lrelu, float* amax, void* stream);

/* strided data-gradient for the stride-2 pyramid convs (gather form):
 * gx[b,ci,iy,ix] = sum_{co,tap : iy = oy*stride + (ty-pad)*dil ...} w[co][ci][tap] * gy[b,co,oy,ox] */
int irr_conv2d_dgrad_strided_f32(const float* gy, const float* w, float* gx,
                                 int B, int Cin, int H, int W, int Cout, int OH, int OW,
                                 int k, int stride, int dil, long gy_bs, long gx_bs, void* stream);

/* ---- bilateral refinement tail -------------------------------------------------------------------
 * RefineFlow / RefineOcc tail (models/irr_modules.py:92-104, 130-139):
 *   wgt_t = softmax_t(-f[b,t,y,x]^2), t = dy*3+dx ;
 *   out[b,c,y,x] = scale_c * sum_t wgt_t * v[b,c,clamp(y+dy-1),clamp(x+dx-1)]      (ReplicationPad2d(1) + Unfold(3x3))
 * scale_c = scale0 for c == 0, scale1 otherwise: folds the to_global rescale applied right after RefineFlow
 * (models/IRR_PWC.py:137-138).  f has 9 channels, v/out have C (2 for flow, 1 for occlusion).
 */
int irr_refine_tail_fwd_f32(const float* f, const float* v, float* out, int B, int C, int H, int W,
                            long f_bs, long v_bs, long out_bs, float scale0, float scale1, void* stream);
/* gf (9 channels, nullable) is overwritten; gv (nullable) is zeroed inside and scatter-added. */
int irr_refine_tail_bwd_f32(const float* f, const float* v, const float* gout, float* gf, float* gv,
                            int B, int C, int H, int W, long f_bs, long v_bs, long gout_bs, long gf_bs, long gv_bs,
                            float scale0, float scale1, void* stream);

/* ---- nearest x2 upsampling ------------------------------------------------------------------------
 * upsample_factor2 (models/irr_modules.py:21-27) for H, W multiples of 64 (the bilinear fallback never fires).
 */
int irr_upsample_nearest2x_fwd_f32(const float* x, float* out, int B, int C, int H, int W,
                                   long x_bs, long out_bs, void* stream);
int irr_upsample_nearest2x_bwd_f32(const float* gout, float* gx, int B, int C, int H, int W,
                                   long gout_bs, long gx_bs, void* stream);

/* ---- multi-scale loss, per-pixel parts (losses.py:8-18, 39-48, 515-577) ------------------------------------
 * The scalar algebra (level weights 0.32..0.0003125, flow/occ balancing) stays on the host side; these entry points
 * do everything that touches pixels.
 *   irr_avgpool_f32      out = scale * s x s mean of in  (== adaptive_avg_pool2d for the integer ratios 1..64), in (BC,h*s,w*s)
 *   irr_epe_sum_fwd      *out += weight * sum_{b,p} || tgt - flow ||_2            flow, tgt: (B,2,h,w)
 *                        (ONE add of a sum formed in a fixed order: block partials in `scratch`, then one finishing block --
 *                        no atomics, bit-reproducible; scratch >= irr_loss_partial_blocks(B, HW) floats)
 *   irr_epe_sum_bwd      gflow = gscale[0]*weight * (flow - tgt)/||.||            (0 where the norm is 0)
 *   irr_f1bal_sums       sums[b][0..3] = { -sum t log(s+eps), -sum (1-t) log(1-s+eps), sum t, sum s },  s = sigmoid(logit)
 *                        (overwritten; fixed summation order; scratch >= 4 * irr_loss_partial_blocks(B, HW) floats, 16-B aligned)
 *   irr_f1bal_value      out[0] = scale * sum_b [ tp/(st+sp+eps) + fn/((N-st)+(N-sp)+eps) ]   (sums from irr_f1bal_sums, N = HW)
 *   irr_f1bal_bwd        glogit = gscale[0]*weight * d/dlogit [ tp/(st+sp+eps) + fn/((N-st)+(N-sp)+eps) ]
 * gscale is a 1-element DEVICE array (the upstream gradient times the balancing weight), so nothing syncs. */
int irr_avgpool_f32(const float* in, float* out, int BC, int h, int w, int s, float scale, void* stream);
/* the general case of losses.py:16-18: out (BC,h,w) = scale * adaptive_avg_pool2d(in (BC,H,W), [h, w]) for level sizes that do
 * not divide the target size (window [floor(o*H/h), ceil((o+1)*H/h)) per axis, as ATen) -- odd pyramid sizes */
int irr_adaptive_avgpool_f32(const float* in, float* out, int BC, int H, int W, int h, int w, float scale, void* stream);
long irr_loss_partial_blocks(int B, long HW);   /* blocks (= partial-sum slots) the forward reductions use for one (B, HW) term */
int irr_epe_sum_fwd_f32(const float* flow, const float* tgt, float* out, int B, int HW, long flow_bs, long tgt_bs,
                        float weight, float* scratch, long scratch_elems, void* stream);
int irr_epe_sum_bwd_f32(const float* flow, const float* tgt, const float* gscale, float* gflow, int B, int HW,
                        long flow_bs, long tgt_bs, long g_bs, float weight, void* stream);
int irr_f1bal_sums_f32(const float* logit, const float* tgt, float* sums, int B, int HW, long l_bs, long t_bs,
                       float* scratch, long scratch_elems, void* stream);
int irr_f1bal_value_f32(const float* sums, float* out, int B, int HW, float scale, void* stream);
int irr_f1bal_bwd_f32(const float* logit, const float* tgt, const float* sums, const float* gscale, float* glogit,
                      int B, int HW, long l_bs, long t_bs, long g_bs, float weight, void* stream);

/* All terms of one kind in ONE launch (the loss has 24 EPE and 24 balanced-F1 terms).  `terms` is a HOST array of nterms
 * (<= IRR_LOSS_MAX_TERMS) records; it is copied into the kernel arguments, nothing is read from it after the call returns.
 *   pred / tgt   the term's prediction (flow (B,2,h,w) or occlusion logits (B,1,h,w)) and its pooled target
 *   grad         (bwd) where the gradient w.r.t. pred goes;  aux: (F1) the term's sums[B][4] (written by fwd, read by bwd; 16-B aligned)
 *   weight       EPE: level weight;  F1: level weight * h*w*0.5  (what irr_f1bal_value_f32 takes as `scale`)
 *   hw, *_bs     pixels per plane and batch strides in elements;  nbx / block0 are filled in by the library
 * fwd:  out[0] += sum over terms of the term's weighted loss (out is NOT zeroed), summed in a FIXED order: every block stores its
 *       partial sum(s) in `scratch` (EPE: sum_t irr_loss_partial_blocks(B_t, hw_t) floats, F1: four times that, 16-B aligned) and one
 *       finishing block adds them in index order -- no atomics, the loss values are bit-reproducible;
 * bwd:  grad_t = gscale[0] * d term / d pred. */
#define IRR_LOSS_MAX_TERMS 32
typedef struct IrrLossTerm {
  const float* pred;
  const float* tgt;
  float* grad;
  float* aux;
  long hw, pred_bs, tgt_bs, grad_bs;
  float weight;
  int B, nbx, block0;
} IrrLossTerm;
int irr_epe_sum_multi_fwd_f32(const void* terms, int nterms, float* out, float* scratch, long scratch_elems, void* stream);
int irr_epe_sum_multi_bwd_f32(const void* terms, int nterms, const float* gscale, void* stream);
int irr_f1bal_multi_fwd_f32(const void* terms, int nterms, float* out, float* scratch, long scratch_elems, void* stream);
int irr_f1bal_multi_bwd_f32(const void* terms, int nterms, const float* gscale, void* stream);

/* ---- channel concatenation in one launch --------------------------------------------------------------
 * Replaces torch.cat on the decoder / upsampler inputs (models/IRR_PWC.py:104-107, 166-167; models/pwc_modules.py:164-168 builds
 * its DenseNet buffer the same way): `parts` is a HOST array of nparts (<= IRR_CAT_MAX_PARTS) records, copied into the kernel
 * arguments.  Part i = (B, channels_i, hw) with dense planes and batch stride src_bs (elements) is written to the channels
 * [sum_{j<i} channels_j, ...) of dst (batch stride dst_bs, channel stride hw); src == NULL writes zeros. */
#define IRR_CAT_MAX_PARTS 8
typedef struct IrrCatPart {
  const float* src;
  long src_bs;
  int channels;
  int reserved;
} IrrCatPart;
int irr_cat_channels_f32(float* dst, long dst_bs, const void* parts, int nparts, int B, long hw, void* stream);
/* the same + *amax = max(*amax, max |value written|): the consumer's fp16x2 input magnitude without a pass over the buffer (ABI 7) */
/* out[b, 0:n) = x[b, 0:n) + y[b, 0:n) for B samples of n contiguous floats with independent batch strides (elements); out may
 * alias x or y.  The sum of a channel-slice view and a dense tensor in the backward passes (models/irr_modules.py:55-56 skip,
 * models/pwc_modules.py:169-170 head) as one coalesced pass instead of ATen's strided-iterator kernel (ABI 7). */
int irr_add_planes_f32(float* out, const float* x, const float* y, int B, long n, long out_bs, long x_bs, long y_bs, void* stream);
int irr_cat_channels_amax_f32(float* dst, long dst_bs, const void* parts, int nparts, int B, long hw, float* amax, void* stream);
/* (ABI 12) ... and chmax[c] = max(chmax[c], max |dst[:, c]| as written) per destination channel (zero-initialised floats, one per channel
 * written by this call; amax nullable): the channel maxima of an assembled decoder input without a pass (irr_conv2d_wgrad_h2_ch). */
int irr_cat_channels_amax_ch_f32(float* dst, long dst_bs, const void* parts, int nparts, int B, long hw, float* amax, float* chmax, void* stream);

/* ---- fused Adam over one flat arena ------------------------------------------------------------------
 * torch.optim.Adam semantics (runtime.py:189; lr 1e-4, weight_decay 4e-4 as L2-in-gradient,
 * scripts/IRR-PWC_flyingChairsOcc.sh:29-31) over n contiguous fp32 elements (16-byte aligned pointers):
 *   g = grad*grad_scale + wd*p ; m = lerp(m, g, 1-b1) ; v = b2*v + (1-b2)*g*g ;
 *   p -= (lr/bias_corr1) * m / (sqrt(v)/sqrt(bias_corr2) + eps)          bias_corr_i = 1 - beta_i^t
 * step_dev (nullable): DEVICE float holding the step count t; when given, the bias corrections are computed from it inside
 * the kernel and bias_corr1/2 are ignored -- a captured launch (hipGraph) then stays correct on every replay.
 * The scalar hyper-parameters are DOUBLES, as in torch.optim.Adam: 1 - beta_i, lr / bias_corr1 and sqrt(bias_corr2) are formed
 * in double and rounded to fp32 once (1.f - 0.999f is 1.3e-5 away from 0.001).
 */
int irr_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n,
                      double lr, double beta1, double beta2, double eps, double weight_decay,
                      double bias_corr1, double bias_corr2, double grad_scale, const float* step_dev, void* stream);

/* ---- on-GPU training augmentation: RandomAffineFlowOcc (augmentations.py:368-653) -----------------------------
 * The random parameters (thetas, mirror signs, crop origin) are sampled by the host exactly as the reference does
 * (augmentations.py:469-517, 66-96, 565-585); these entry points do the per-pixel work, computing only the crop window
 * [y0, y0+OH) x [x0, x0+OW) of the H x W augmented frame (OH=H, OW=W, y0=x0=0 when there is no crop).
 * Per batch element, ``inv`` holds (b1,b2,b4,b5,a3,a6) of the inverted map and ``theta`` the six affine parameters
 * (a1..a6) in normalised [-1,1] coordinates (transform_coords :415-440, inverse_transform_coords :391-413).
 *
 *   irr_affine_warp_f32      transform_image (:519-523) = transform_coords + Interp2(clamp=False)
 *                            (utils/interpolation.py:82-141: floor/clamped neighbours, zero where the query leaves the
 *                            frame).  noise != NULL (standard-normal samples, dense B x C x OH x OW) additionally applies
 *                            clamp(v + noise_std*n, 0, 1) (augmentations.py:630-637).  src (B,C,H,W), dst (B,C,OH,OW).
 *   irr_affine_flow_occ_f32  transform_flow (:525-548) of ``flow`` under (theta_a -> theta_b), resampled with inv_a, and,
 *                            when occ/occ_out are given (both or neither), transform_image of the 1-channel ``occ`` fused
 *                            with check_out_of_bound (:550-563) evaluated in the cropped frame:
 *                            occ_out = clamp(occ' + [x+u', y+v' outside OW x OH], 0, 1).
 */
int irr_affine_warp_f32(const float* src, float* dst, const float* inv, const float* noise, float noise_std,
                        int B, int C, int H, int W, int OH, int OW, int y0, int x0, long src_bs, long dst_bs,
                        void* stream);
int irr_affine_flow_occ_f32(const float* flow, const float* occ, float* flow_out, float* occ_out,
                            const float* inv_a, const float* theta_a, const float* theta_b, int B, int H, int W,
                            int OH, int OW, int y0, int x0, long flow_bs, long occ_bs, long flow_out_bs,
                            long occ_out_bs, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IRR_HIP_H */
