/* plshts.h -- C ABI of the MI355X (gfx950) HEALPix spherical-harmonic-transform engine.
 *
 * Drop-in boundary for the SHT seam of carronj/plancklens.  The reference reaches its SHTs through
 * Python functions with healpy's signatures (plancklens/shts.py:12-35, and direct hp.* calls at
 * qest.py:259,280,464,504,514,530, filt/filt_simple.py:399,404, qcinv/opfilt_tt.py:34,187,189,
 * qcinv/opfilt_pp.py:260,265,314, qcinv/opfilt_tp.py:23,276,281).  A ctypes module
 * (plancklens_amd/_lib.py) binds the entry points below and re-creates those signatures
 * (plancklens_amd/shts.py); INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Data conventions (healpy):
 *   alm : complex128 as interleaved (re, im) doubles, m-major triangular, idx(l,m) = m(2 lmax+1-m)/2 + l,
 *         mmax = lmax, size nalm = (lmax+1)(lmax+2)/2; spin > 0 takes [G | C] = 2 * nalm complex.
 *   map : float64 RING-ordered, npix = 12 nside^2; spin > 0 takes [Q | U] = 2 * npix doubles.
 *   map2alm uses uniform pixel weights 4 pi / npix and no iterations (every reference call passes iter=0).
 *
 * All pointers are plain device or host pointers (flag PL_HOST / PL_DEVICE); no framework types.
 * Every call is stream-ordered on `stream` (a hipStream_t passed as void*, NULL = default stream) and
 * asynchronous with device pointers; with host pointers it synchronises before returning.
 * Return value: 0 on success, non-zero on error (message from pl_last_error()).
 * A plan is bound to the device that was current when it was created and is not thread-safe.
 */
#ifndef PLSHTS_H
#define PLSHTS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pl_plan pl_plan;

#define PL_HOST 0   /* alm / map / fl pointers are host memory   */
#define PL_DEVICE 1 /* alm / map / fl pointers are device memory */

/* ABI version of this header (bumped on incompatible change). */
int pl_version(void);
const char *pl_last_error(void);
/* Number of visible HIP devices, or -1 (with pl_last_error set) if the runtime cannot be initialised. */
int pl_device_count(void);

/* A second execution context on the same tables: own workspaces and side streams, geometry / recursion / FFT tables
 * shared with (and owned by) `plan`.  Transforms issued on different forks and different streams may run concurrently --
 * the ring-FFT kernels of one hide under the FMA-bound Legendre kernels of another.  Destroy forks before their parent. */
int pl_plan_fork(pl_plan *plan, pl_plan **fork);

/* Plan: HEALPix ring geometry, recursion tables and FFT tables for one (nside, lmax, mmax = lmax).
 * Replaces what healpy / libsharp build internally per call (shts.py:13,18,23,28 build a geometry per call). */
int pl_plan_create(int nside, int lmax, pl_plan **plan);
/* Plan options: the ring-FFT routing decisions taken at plan creation, as explicit arguments (the library reads no environment
 * variable at plan creation; two plans of one process differ only where the caller said so).  -1 = the measured default. */
typedef struct pl_plan_opts {
    int fft_legacy;    /* 1: every ring pair through the generic LDS-resident ring-FFT kernel (what the small-size oracle tests pin) */
    int fft_split_min; /* smallest half-size for split Bluestein rings (default 512); 0: never split */
    int fft_nyq_min;   /* shortest power-of-two sub-DFT whose order-n/2 rings use the register classes (default 2048); 0: never */
    int fft_min_fast;  /* a plan whose register classes would hold under 1/d of its ring pairs runs every ring in the generic kernel
                          (default d = 8); 0: never */
    int fft_generic_nside; /* grids up to this nside run every ring in the generic kernel: one launch per stage instead of a dozen
                          latency-bound class kernels on side streams (default 512: measured 1.2-4x faster stages at nside 256 / 512,
                          1.5-2x slower at 1024); 0: never */
    int seed_tables;   /* 1 (default): the plan keeps, per kernel family, the Legendre recursion state of every (m, ring pair) at the step
                          where the family's kernels would stop recursing without accumulating (no ring of the wavefront has reached the
                          activation threshold yet: ~10 % of all recursion steps, at the latency of their dependent FMA chains), made once
                          by the same arithmetic -- transforms start from it, results bit-identical; ~40 B per (m, ring pair) and spin
                          (0.34 GB for spin 0 + 0.67 GB per spin s at nside = lmax = 2048; x 4 at 4096).  0: none, every launch recurses from l = m */
} pl_plan_opts;
/* pl_plan_create (rank 0 of 1) / pl_plan_create_shard with options; opts NULL = defaults. */
int pl_plan_create_opts(int nside, int lmax, int rank, int nranks, const pl_plan_opts *opts, pl_plan **plan);
int pl_plan_destroy(pl_plan *plan);
int64_t pl_plan_npix(const pl_plan *plan);
int64_t pl_plan_nalm(const pl_plan *plan);
int64_t pl_plan_bytes(const pl_plan *plan); /* device bytes held by the plan */
/* The plan's i-th side stream (a hipStream_t; NULL when i is out of range): the streams the classes of a ring-FFT stage are spread over
 * between a fork and a join on the caller's stream.  For tools that look at how streams fall onto the hardware queues. */
void *pl_plan_side_stream(const pl_plan *plan, int i);
/* Recursion steps the Legendre kernels of one family (fam 0: synthesis, 1: analysis) of the plan execute per launch and input, read from
 * the plan's seed table (pl_plan_opts.seed_tables): (l, ring pair) steps for spin >= 1 (12 FMA each), two-l steps for spin 0 (6 FMA each);
 * every ring-pair slot of a wavefront that runs is counted.  -1 without a table.  For measurement tools (bench.py's executed flops). */
int64_t pl_plan_executed_steps(pl_plan *plan, int spin, int fam);
/* The same count restricted to the slots that hold a ring which keeps the order (ring pair < npairs, m <= its pruning limit): the steps
 * whose results are used -- the difference is lanes of running wavefronts that ride along on pruned rings or padding. */
int64_t pl_plan_useful_steps(pl_plan *plan, int spin, int fam);

/* One transform over several GPUs ("m-blocks shard across the GPUs", BASELINE.json north_star; the reference's only parallelism
 * inside a transform is the third-party library's threads, shts.py:10).  Shard `rank` of `nranks`:
 *   - the Legendre-stage calls on this plan (pl_legendre_synth / _anal and the transforms built on them) cover the m-groups
 *     rank, rank + nranks, ... -- an m-group is 4 consecutive orders m, the unit of one workgroup;
 *   - its ring-FFT calls (pl_phase2map / pl_map2phase) cover the ring pairs rank, rank + nranks, ... (interleaved: every ring-length
 *     class stays balanced), i.e. read / write only the pixels of those rings;
 *   - between the stages the ranks exchange phase slices: pl_phase_pack gathers the sub-grid (ring pairs pair0 + j pair_stride,
 *     m-groups mg0 + k mg_stride) of a phase array into the contiguous buffer [j][component][k][4 orders][4 doubles] that one rank
 *     sends another, pl_phase_unpack scatters a received buffer (one all-to-all per transform: plancklens_amd/parallel.py);
 *   - pl_alm_keep_mgroups zeroes the alm entries outside a rank's m-groups, so that the analysis results of the ranks add up
 *     exactly (every entry is non-zero on one rank only).
 * Results equal the single-plan transform up to the order of the ring-group partial sums of the analysis (1e-15 relative). */
int pl_plan_create_shard(int nside, int lmax, int rank, int nranks, pl_plan **plan);
int pl_phase_pack(pl_plan *plan, int ncomp, const double *phase, double *buf, int pair0, int pair_stride, int mg0, int mg_stride, void *stream);
int pl_phase_unpack(pl_plan *plan, int ncomp, double *phase, const double *buf, int pair0, int pair_stride, int mg0, int mg_stride, void *stream);
int64_t pl_phase_pack_doubles(const pl_plan *plan, int ncomp, int pair0, int pair_stride, int mg0, int mg_stride);
int pl_alm_keep_mgroups(int lmax, int nb, double *alm, int mg0, int mg_stride, void *stream);
/* The pixels of the ring pairs pair0, pair0 + pair_stride, ... of an ncomp-component map (components npix apart) packed as
 * [component][pair: north ring, south ring], pl_map_pack_doubles doubles per component, and back: a rank's own rings for the all-gather
 * that completes a sharded synthesis (1 / R of the map instead of an all-reduce of zero-padded maps; plancklens_amd/parallel.py). */
int64_t pl_map_pack_doubles(const pl_plan *plan, int pair0, int pair_stride);
int pl_map_pack_rings(pl_plan *plan, int ncomp, const double *map, double *buf, int pair0, int pair_stride, void *stream);
int pl_map_unpack_rings(pl_plan *plan, int ncomp, double *map, const double *buf, int pair0, int pair_stride, void *stream);

/* shts.alm2map (shts.py:12-15) / shts.alm2map_spin (shts.py:22-24).  spin = 0: alm -> map;
 * spin = 1,2,3: [G|C] -> [Q|U].  If fl != NULL (length lmax + 1) the alm are multiplied by fl_l on the
 * fly (hp.almxfl fused; qest.py:463,502-503,592).  Inputs are not modified. */
int pl_alm2map(pl_plan *plan, int spin, const double *alm, double *map, const double *fl, int where, void *stream);

/* pl_alm2map with spin >= 1 for an input whose curl component is identically zero (the gradient legs of the temperature
 * lensing estimator, qest.py:453-464,566-595, hand hp.alm2map_spin a zero C array): almG holds the nalm gradient
 * coefficients only; map receives both components.  Same result as pl_alm2map with C = 0 at 2/3 of the Legendre work. */
int pl_alm2map_grad(pl_plan *plan, int spin, const double *almG, double *map, const double *fl, int where, void *stream);

/* shts.map2alm(iter=0) (shts.py:16-20) / shts.map2alm_spin (shts.py:26-30).  If fl != NULL the
 * result is multiplied by fl_l (hp.almxfl fused; filt_simple.py:400,405-406, qest.py:261-262). */
/* Two spin-s syntheses on ONE recursion: a general input (G, C) with filter fl and a gradient-only input G2 (curl = 0)
 * with filter fl2 -- the spin-1 legs of the minimum-variance estimator, alm2map_spin((E^WF, B^WF) w^1) and
 * alm2map_spin((-sqrt(l(l+1)) T^WF, 0)) (qest.py:453-464,597-638).  maps4_dev = [Q | U | Q2 | U2], 4 npix doubles.
 * Device pointers only, asynchronous on `stream`.  16 + 4 instead of 12 + 4 + 8 + 4 FMAs per recursion step. */
int pl_alm2map_pair(pl_plan *plan, int spin, const double *alm_gc_dev, const double *fl_dev, const double *alm_g2_dev, const double *fl2_dev,
                    double *maps4_dev, void *stream);
/* Two gradient-only spin-s syntheses (curl input identically zero: pl_alm2map_grad) on ONE Legendre recursion -- the gradient legs
 * alm2map_spin([-sqrt(l(l+1)) T^WF_lm, 0], 1) of the temperature estimator (plancklens/qest.py:453-464) for two simulations: 12 instead
 * of 2 x 8 FMAs per (l, m, ring pair).  maps4: (re1, im1, re2, im2); each pair equals pl_alm2map_grad of its input bit for bit. */
int pl_alm2map_grad_pair(pl_plan *plan, int spin, const double *alm_g1, const double *fl1, const double *alm_g2, const double *fl2, double *maps4,
                         void *stream);
/* The same spin-s synthesis of TWO general inputs (two simulations) on ONE recursion: 4 + 8 + 8 instead of 2 x (4 + 8) FMAs per
 * recursion step (batching independent maps).  alm_gc_k_dev = [G_k | C_k], one filter fl for both;
 * maps4_dev = [Q1 | U1 | Q2 | U2].  Device pointers only, asynchronous on `stream`.  Bit-identical to two pl_alm2map calls.
 * spin 0 (round 6): alm_gc_k_dev = one scalar alm array each, maps4_dev = [T1 | T2] (two rows); on grids of nside >= 1024 the two inputs share the
 * recursion (10 instead of 2 x 6 FMAs per two-l step and ring pair), elsewhere they are two workgroup sets of one launch; bit-identical to two pl_alm2map calls. */
int pl_alm2map_batch2(pl_plan *plan, int spin, const double *alm_gc_1_dev, const double *alm_gc_2_dev, const double *fl_dev, double *maps4_dev,
                      void *stream);
int pl_map2alm(pl_plan *plan, int spin, const double *map, double *alm, const double *fl, int where, void *stream);
/* pl_map2alm on device maps whose ADDRESSES are read, when the kernels run, from a table in device memory (one entry per component: 1 for spin 0, 2 for
 * spin s; each an npix map anywhere in device memory).  A launch captured into a HIP graph can be replayed on other input maps by rewriting the table --
 * no copy into fixed input slots (qest.library._pair_graph) -- and (Q, U) need not be the rows of one array.  Device pointers only; same kernels and
 * arithmetic as pl_map2alm (hp.map2alm(_spin)(..., iter=0), shts.py:16-30), bit-identical results. */
int pl_map2alm_ind(pl_plan *plan, int spin, const double *const *maps_ind_dev, double *alm_dev, const double *fl_dev, void *stream);
/* Writes n <= 8 device addresses (a HOST array, read before the call returns: passed to the kernel by value) into the table `table_dev` that
 * pl_map2alm_ind reads, by a one-workgroup kernel on `stream` (ordered like any other launch; no host buffer to keep alive). */
int pl_store_addresses(int n, const unsigned long long *addrs, unsigned long long *table_dev, void *stream);

/* Stage-level entry points: tests, stage timings, and callers that pipeline independent transforms (the Legendre stage of
 * one on the caller's stream while the ring FFTs of another run on a second stream and a fork of the plan).
 * phase buffer: [npairs][ncomp][mstride][4] doubles = (F_north re, im, F_south re, im): the orders of one component of
 * one ring pair are contiguous. */
int64_t pl_plan_phase_doubles(const pl_plan *plan, int spin);
int pl_legendre_synth(pl_plan *plan, int spin, const double *alm_dev, const double *fl_dev, double *phase_dev, void *stream);
int pl_legendre_synth_grad(pl_plan *plan, int spin, const double *almG_dev, const double *fl_dev, double *phase_dev, void *stream); /* see pl_alm2map_grad */
int pl_legendre_anal(pl_plan *plan, int spin, const double *phase_dev, double *alm_dev, const double *fl_dev, void *stream);
int pl_phase2map(pl_plan *plan, int spin, const double *phase_dev, double *map_dev, void *stream);
int pl_map2phase(pl_plan *plan, int spin, const double *map_dev, double *phase_dev, void *stream);

/* Per-stage timing with HIP events recorded on the caller's stream around the dominant kernels (used by
 * bench.py for the roofline numbers).  Kinds: 0 Legendre synthesis spin 0, 1 Legendre synthesis spin s,
 * 2 Legendre analysis spin 0 (+ reduction), 3 Legendre analysis spin s (+ reduction), 4 ring FFT synthesis,
 * 5 ring FFT analysis, 6 gradient-only Legendre synthesis spin s (pl_legendre_synth_grad: 16 instead of 24 flop
 * per recursion step, kept apart so that kind 1 prices full launches only), 7 paired Legendre synthesis (pl_alm2map_pair:
 * 32 flop per step for two transforms), 8 batched Legendre synthesis (pl_alm2map_batch2: 40 flop per step for two maps), 9 two scalar
 * syntheses on one recursion (pl_alm2map_batch2 with spin 0 / an even block of pl_cg_fwd_tt_b on a fine grid: 20 instead of 2 x 12 flop per two-l step).  pl_profile_read synchronises the recorded events, returns summed milliseconds and
 * launch counts per kind (arrays of PL_PROFILE_KINDS entries) and resets the record. */
#define PL_PROFILE_KINDS 10
int pl_profile_enable(pl_plan *plan, int on);
int pl_profile_read(pl_plan *plan, double *ms_sum, int64_t *counts);

/* Harmonic-space helpers on device arrays (hp.almxfl: 147 call sites; hp.alm2cl / dot_op:
 * opfilt_tt.py:43-51, opfilt_pp.py:27-34, qecl.py:147-148; utils.alm_copy: utils.py:19-35). */
int pl_almxfl(int lmax, const double *alm_in, const double *fl, int nfl, double *alm_out, void *stream);
int pl_alm2cl(int lmax, const double *alm_a, const double *alm_b, double *cl_out /* lmax+1, device */, void *stream);
int pl_alm_copy(int lmax_in, const double *alm_in, int lmax_out, double *alm_out, void *stream);
/* out = a * x + y on n doubles (cd_solve.py:75,86,102 axpy on alm arrays), all device pointers. */
int pl_axpy(int64_t n, double a, const double *x, const double *y, double *out, void *stream);

/* CG vector primitives with device-resident scalars: one launch each, no host synchronisation (they are what the
 * coarse multigrid levels consist of once their SHTs are small).  A scalar product lives in device memory as
 * PL_DOT_PARTS partial sums (its value is their sum in index order): no second launch, no atomics, bit-reproducible.
 * pl_alm_dot: parts_dev[0 .. PL_DOT_PARTS) (+)= partial sums of sum_{l >= lmin} sum_m w_m Re(a_lm conj(b_lm)), w_0 = 1,
 *   w_m = 2 -- the scalar product sum_l (2l + 1) C_l^{ab} of opfilt_tt.py:43-51 (lmin = 2: opfilt_pp.py:27-34).
 * pl_axpy_dev: y += sign * num / den * x on n doubles, num and den given as such partial sums (den NULL: 1) -- the
 *   updates of cd_solve.py:75,86,102 with alpha = delta / dTAd formed on the device.
 * pl_alm_splice: out = alm_lo for l <= lsplit, alm_hi above, band-limit lmax_hi (util_alm.py:8-24).
 * pl_almxfl_add: out = a + f_l b (out may be a) -- N-part + S^-1 x of fwd_op (opfilt_tt.py:67-73). */
#define PL_DOT_PARTS 64
int pl_alm_dot(int lmax, int lmin, const double *a, const double *b, int accumulate, double *parts_dev, void *stream);
int pl_axpy_dev(int64_t n, const double *num_parts_dev, const double *den_parts_dev, double sign, const double *x, double *y, void *stream);
int pl_alm_splice(int lmax_lo, const double *alm_lo, int lmax_hi, const double *alm_hi, int lsplit, double *out, void *stream);
/* the same with the high part multiplied by fl_hi[l] (lmax_hi + 1 entries, device): pre_op_split with a diagonal high-l
 * preconditioner (multigrid.py:163-182, opfilt_tt.py:76-93) in one launch */
int pl_alm_splice_fl(int lmax_lo, const double *alm_lo, int lmax_hi, const double *alm_hi, const double *fl_hi, int lsplit, double *out,
                     void *stream);
/* Scalar products and the updates they scale in ONE launch, over nf <= 3 fields (host arrays of nf device pointers / band-limits):
 *   parts1 = <a, b1>, parts2 = <a, b2> (b2 given), left in parts*_dev as pl_alm_dot leaves them (summed over the fields);
 *   c = parts2 / parts1 (b2 given) or parts1 / den_parts_dev (den given; exactly one of b2, den_parts_dev);
 *   y1 += sign1 c x1 and, when y2 is given, y2 += sign2 c x2 (signs +-1).
 * With (a, b1, b2, y1, x1, y2, x2) = (d, Ad, r, x, d, r, Ad), signs (+1, -1) this is one conjugate-directions update
 * (cd_solve.py:66-84); with (s, Ad', -, s, d') and den = d'^t A d', sign -1 the re-orthogonalisation (cd_solve.py:96-103).
 * Same arithmetic as pl_alm_dot + pl_axpy_dev (bit-identical).  barrier_dev NULL: two launches (all products, then all updates).
 * barrier_dev given: ONE launch, a grid-wide barrier separating the products from the updates -- 4 zero-initialised 32-bit words in
 * device memory owned by the caller (barrier_dev[2] != 0 afterwards reports a barrier that timed out, results invalid; launches on
 * one stream only may share the words).  On MI355X the barrier costs what the kernel boundary costs: the two forms run equally fast. */
int pl_cg_dot_axpy(int nf, const int *lmax, int lmin, const double *const *a, const double *const *b1, const double *const *b2,
                   double *parts1_dev, double *parts2_dev, const double *den_parts_dev, double *const *y1, const double *const *x1,
                   double sign1, double *const *y2, const double *const *x2, double sign2, unsigned *barrier_dev, void *stream);

/* out[k] = fl[2k] alm[2k] (+ fl[2k+1] alm[2k+1]), k < nout <= 2 outputs of nterm[k] = 1 or 2 terms each, all of band-limit lmax with filters of
 * lmax + 1 entries: the Wiener-filtered legs X^WF = C^XX Xb (+ C^TE Yb) of the quadratic estimators (plancklens/qest.py:566-638, where they
 * are three hp.almxfl calls and an addition) for both components of a spin transform in one launch; rounded as pl_almxfl then pl_almxfl_add. */
int pl_alm_lincomb(int lmax, int nout, const int *nterm, const double *const *alm, const double *const *fl, double *const *out, void *stream);
/* Inverse-noise weighting with template marginalisation (alm_filter_ninv.apply_map, opfilt_tt.py:196-205) in two
 * launches: tmap <- n_inv tmap - sum_k rmat[k] c_k, c_k = sum_i pmat[k][i] n_inv[i] tmap[i], with pmat (nmodes x npix,
 * the template modes) and rmat = (P^t N^-1 P)^-1 (pmat . n_inv) (nmodes x npix), all device arrays; scratch_dev:
 * PL_TEMPLATE_MAX_MODES * 256 doubles.  Bit-reproducible. */
#define PL_TEMPLATE_MAX_MODES 16
int pl_template_project(int64_t npix, int nmodes, double *tmap, const double *n_inv, const double *pmat, const double *rmat,
                        double *scratch_dev, void *stream);
int pl_almxfl_add(int lmax, const double *a, const double *b, const double *fl, int nfl, double *out, void *stream);

/* The temperature CG operator x -> S^-1 x + B^t Y^t N^-1 Y B x (fwd_op.calc, opfilt_tt.py:67-73, with apply_alm :184-194 and the
 * template-marginalised apply_map :196-205 inside) as one call on the plan's grid:
 *   alm_out = fl_out * Y^t [n_inv t - sum_k rmat[k] c_k],  t = Y (fl_in * alm_in),  c_k = sum_i pmat[k][i] n_inv[i] t[i],
 * plus fl_add * alm_add when alm_add is given (alm_out may be neither input).  nmodes = 0: plain N^-1 weighting (pmat, rmat,
 * scratch_dev unused).  On grids whose rings all run in the generic ring-FFT kernel and nmodes <= 4 the weighting and the projection
 * ride in the two FFT launches; otherwise they are the pl_template_project launches.  All arrays in device memory; fl_* have
 * lmax + 1 entries (null: 1); scratch_dev as pl_template_project.  Bit-reproducible. */
int pl_cg_fwd_tt(pl_plan *plan, const double *alm_in, const double *fl_in, const double *n_inv, int nmodes, const double *pmat,
                 const double *rmat, double *scratch_dev, const double *alm_add, const double *fl_add, double *alm_out, const double *fl_out,
                 void *stream);
/* The polarization counterpart (fwd_op.calc, opfilt_pp.py:69-78 with apply_alm :190-205 and the one-map apply_map :207-215):
 *   (E, B)_out = fl_out * Y2^t [n_inv * Y2 (fl_in * (E, B)_in)]  +  (fl_add_e * E_add, fl_add_b * B_add),
 * E and B as separate arrays of nalm complex numbers, n_inv the single inverse-noise map shared by Q and U; the weighting rides in the
 * synthesis-side ring-FFT launches, the add terms in the post-processing of the analysis.  elm_add / blm_add may both be null. */
int pl_cg_fwd_pp(pl_plan *plan, const double *elm_in, const double *blm_in, const double *fl_in, const double *n_inv, const double *elm_add,
                 const double *blm_add, const double *fl_add_e, const double *fl_add_b, double *elm_out, double *blm_out,
                 const double *fl_out, void *stream);

/* ---- Batched forms: nb right-hand sides through every launch (block vectors of the conjugate-gradient filter) -------------------
 * The reference filters its Monte-Carlo simulations one at a time (examples/run_qlms.py:57-62 -> filt_cinv.py:196-203,275-289 ->
 * multigrid.py:56-75 -> cd_solve.py:35-107).  On the GPU the coarse multigrid levels of a solve are chains of dependent launches of a
 * few microseconds each; nb simulations that share the noise model are therefore solved together: every launch of the solve carries
 * all of them, each with its own scalar products and step lengths.  Convention: every alm / map argument holds nb arrays back to
 * back ([nb][nalm] complex, [nb][npix]); l-filters, inverse-noise maps and template matrices are shared by the batch; partial sums of
 * scalar products are [nb][PL_DOT_PARTS]; scratch_dev of the template projection is nb x PL_TEMPLATE_MAX_MODES x 256 doubles.
 * Every entry of a batch is computed with exactly the arithmetic of the un-batched call (bit-identical results). */
#define PL_MAX_BATCH 64
int pl_almxfl_b(int lmax, int nb, const double *alm_in, const double *fl, int nfl, double *alm_out, void *stream);
int pl_alm_copy_b(int lmax_in, int nb, const double *alm_in, int lmax_out, double *alm_out, void *stream);
/* fl_hi may be NULL (pl_alm_splice) or lmax_hi + 1 entries (pl_alm_splice_fl) */
int pl_alm_splice_b(int lmax_lo, int nb, const double *alm_lo, int lmax_hi, const double *alm_hi, const double *fl_hi, int lsplit, double *out,
                    void *stream);
int pl_almxfl_add_b(int lmax, int nb, const double *a, const double *b, const double *fl, int nfl, double *out, void *stream);
int pl_alm_dot_b(int lmax, int lmin, int nb, const double *a, const double *b, int accumulate, double *parts_dev, void *stream);
int pl_axpy_dev_b(int64_t n, int nb, const double *num_parts_dev, const double *den_parts_dev, double sign, const double *x, double *y, void *stream);
/* pl_cg_dot_axpy for nb entries (the host pointer arrays give the first entry of each field; always the two-launch form).
 * active_dev (may be NULL): nb doubles, 1 or 0 -- the step length of entry b is multiplied by active_dev[b], so that an entry whose
 * solve has met its stopping criterion keeps its vectors while the others go on (cd_monitors.py:28-41 applied per simulation). */
int pl_cg_dot_axpy_b(int nb, int nf, const int *lmax, int lmin, const double *const *a, const double *const *b1, const double *const *b2,
                     double *parts1_dev, double *parts2_dev, const double *den_parts_dev, double *const *y1, const double *const *x1,
                     double sign1, double *const *y2, const double *const *x2, double sign2, const double *active_dev, void *stream);
/* Scalar products formed by the kernel that writes the vector, instead of a scalar-product launch of their own (the conjugate-directions step of
 * plancklens/qcinv/cd_solve.py:66-84: dTAd = <d, q>, delta = <d, r> with q = fwd_op(d)):
 * pl_plan_arm_post_dots is one shot: the next pl_cg_fwd_tt* (nf = 1) / pl_cg_fwd_pp* (nf = 2) call on this plan also leaves, per batch entry,
 * pl_post_dots_count(plan) partial sums of <d, q> in pre1_dev and of <d, r> in pre2_dev (q its result; weights of pl_alm_dot, entries l < lmin
 * excluded; d, r: nf device arrays laid out as the operator's output); any other analysis call on the plan (pl_map2alm, pl_legendre_anal) cancels a
 * pending request.
 * pl_cg_axpy_pre_b: the updates of pl_cg_dot_axpy_b from such partial sums (npre per batch entry, added in a fixed order by every workgroup):
 * den_parts_dev given: y1 += sign1 sum(pre1) / sum(den) x1; else y1 += sign1 sum(pre2) / sum(pre1) x1 and (y2 non-NULL) y2 += sign2 (the same) x2.
 * parts1_dev / parts2_dev (may be NULL) receive the totals of pre1 / pre2 in the PL_DOT_PARTS-entry form of pl_alm_dot.  y1_assign != 0: y1 = sign1 (...) x1
 * (y1 is written, not read: the first step of a solve that starts from zero needs no zero-filled solution vector). */
int pl_post_dots_count(pl_plan *plan);
int pl_plan_arm_post_dots(pl_plan *plan, int nf, const double *const *d, const double *const *r, int lmin, double *pre1_dev, double *pre2_dev);
int pl_cg_axpy_pre_b(int nb, int nf, const int *lmax, int npre, const double *pre1_dev, const double *pre2_dev, const double *den_parts_dev,
                     double *parts1_dev, double *parts2_dev, double *const *y1, const double *const *x1, double sign1, double *const *y2,
                     const double *const *x2, double sign2, const double *active_dev, int y1_assign, void *stream);
int pl_template_project_b(int64_t npix, int nmodes, int nb, double *tmap, const double *n_inv, const double *pmat, const double *rmat,
                          double *scratch_dev, void *stream);
/* y_b -= rmat^t (pmat x_b), b < nb, vectors of n doubles, pmat / rmat (nmodes, n): the template projection of the CG operators applied
 * in harmonic space -- B^t Y^t [N^-1 - N^-1 T (T^t N^-1 T)^-1 T^t N^-1] Y B x = B^t Y^t N^-1 Y B x - V (T^t N^-1 T)^-1 V^t x with
 * V = B^t Y^t N^-1 T (reference: the bracket in pixel space, plancklens/qcinv/opfilt_tt.py:196-205, opfilt_pp.py:272-303).  x is only
 * read.  scratch_dev as in pl_template_project_b.  Bit-reproducible; every entry equals the nb = 1 call. */
int pl_lowrank_update_b(int64_t n, int nmodes, int nb, const double *x, double *y, const double *pmat, const double *rmat, double *scratch_dev,
                        void *stream);
int pl_cg_fwd_tt_b(pl_plan *plan, int nb, const double *alm_in, const double *fl_in, const double *n_inv, int nmodes, const double *pmat,
                   const double *rmat, double *scratch_dev, const double *alm_add, const double *fl_add, double *alm_out, const double *fl_out,
                   void *stream);
/* pl_cg_fwd_tt_b with plain weighting followed by the harmonic-space template projection alm_out -= hrm^t (hpm alm_in) (hpm, hrm:
 * (nmodes, 2 nalm) real device matrices, pl_lowrank_update_b): one call, the coefficient pass on a side stream of the plan beside the
 * transforms.  alm_out must not alias alm_in. */
int pl_cg_fwd_tt_lr_b(pl_plan *plan, int nb, const double *alm_in, const double *fl_in, const double *n_inv, int nmodes, const double *hpm,
                      const double *hrm, double *scratch_dev, const double *alm_add, const double *fl_add, double *alm_out, const double *fl_out,
                      void *stream);
int pl_cg_fwd_pp_b(pl_plan *plan, int nb, const double *elm_in, const double *blm_in, const double *fl_in, const double *n_inv,
                   const double *elm_add, const double *blm_add, const double *fl_add_e, const double *fl_add_b, double *elm_out,
                   double *blm_out, const double *fl_out, void *stream);
/* Monopole + dipole marginalisation with the templates (1, x, y, z of the pixel centres: template_removal.py:116-150) evaluated from
 * the ring geometry of the plan instead of read as stored maps -- 4 passes over the map instead of 11, three launches:
 *   tmap <- n_inv tmap - n_inv (d_0 + d_1 x + d_2 y + d_3 z),  d = pinv c,  c_k = sum_i (1, x, y, z)_k,i n_inv_i tmap_i,
 * for nb maps back to back; pinv_dev = (P^t N^-1 P)^-1 (4 x 4, row-major, device memory); scratch_dev:
 * pl_template_md_scratch_doubles(plan, nb) doubles.  pl_cg_fwd_tt_md_b is pl_cg_fwd_tt_b with this projection (on every grid: the
 * stored-map form folded into the ring-FFT launches is what the all-generic coarse grids use, pl_plan_fft_all_generic). */
int pl_template_project_md_b(pl_plan *plan, int nb, double *tmap, const double *n_inv, const double *pinv_dev, double *scratch_dev, void *stream);
int64_t pl_template_md_scratch_doubles(const pl_plan *plan, int nb);
int pl_plan_fft_all_generic(const pl_plan *plan);
int pl_cg_fwd_tt_md_b(pl_plan *plan, int nb, const double *alm_in, const double *fl_in, const double *n_inv, const double *pinv_dev,
                      double *scratch_dev, const double *alm_add, const double *fl_add, double *alm_out, const double *fl_out, void *stream);
/* pl_cg_fwd_pp_b for a polarization noise model with a QU cross term (three maps QQ, QU, UU; opfilt_pp.py:295-300): the weighting is
 * one pass of pl_map_qu_weight between the two ring-FFT stages.  nb = 1 for a single right-hand side. */
int pl_cg_fwd_pp_qu_b(pl_plan *plan, int nb, const double *elm_in, const double *blm_in, const double *fl_in, const double *n_qq,
                      const double *n_qu, const double *n_uu, const double *elm_add, const double *blm_add, const double *fl_add_e,
                      const double *fl_add_b, double *elm_out, double *blm_out, const double *fl_out, void *stream);
/* Y[b] = A X[b], b < nb: x [nb][ncols], y [nb][nrows] -- the dense preconditioner applied to a block vector (dense.py:118-119); the
 * matrix is read once for the whole batch. */
int pl_gemv_b(int nrows, int ncols, int64_t lda, const double *A, int nb, const double *x, double *y, void *stream);

/* y = A x, A row-major nrows x ncols with leading dimension lda, all device arrays (x and y must not overlap): the dense
 * low-l preconditioner of the CG chains applied as one mat-vec (dense.py:118-119,201-202,284-285), and the template
 * coefficient products of the joint filter.  One wavefront per row, fixed summation tree (bit-reproducible). */
int pl_gemv(int nrows, int ncols, int64_t lda, const double *A, const double *x, double *y, void *stream);
/* pre_op_split (multigrid.py:163-182) with the dense block below lmax_lo and a diagonal preconditioner above it, one launch, for nf = 1
 * (temperature) or 2 (E, B) fields given as arrays of nf pointers: alm_out[f] (band-limit lmax_hi) = [rows of field f of A x, x = the entries
 * l <= lmax_lo of [alm_hi[0] | alm_hi[1]] | fl_hi[f][l] alm_hi[f] above]; A: pre_op_dense's matrix on the interleaved (re, im) view of the lmax_lo
 * layout(s), map_dev: nalm(lmax_lo) int32 positions of those entries in the lmax_hi layout.  Bit-identical to pl_alm_copy + pl_gemv + pl_alm_splice_fl. */
int pl_gemv_split(int nf, int lmax_lo, int lmax_hi, int64_t lda, const double *A, const double *const *alm_hi, const int *map_dev,
                  const double *const *fl_hi, double *const *alm_out, void *stream);

/* The preconditioner kernels that write the new search direction s of cd_solve.py:93-103 also form <s, q'> (q': the operator applied to the
 * previous direction), so that the re-orthogonalisation needs no scalar-product launch of its own (pl_cg_axpy_pre_b takes the partial sums):
 * pl_gemv_split_dot = pl_gemv_split + pl_gemv_split_dot_count(nf, lmax_lo, lmax_hi) partial sums of sum_f <alm_out[f], q[f]> in pre_dev;
 * pl_alm_splice_dot_b = pl_alm_splice_b (fl_hi may be NULL) + pl_alm_splice_dot_count(lmax_hi) partial sums per batch entry of <out, q>.
 * Weights of pl_alm_dot, entries l < lmin excluded; q laid out as the output. */
int pl_gemv_split_dot_count(int nf, int lmax_lo, int lmax_hi);
int pl_alm_splice_dot_count(int lmax_hi);
int pl_gemv_split_dot(int nf, int lmax_lo, int lmax_hi, int64_t lda, const double *A, const double *const *alm_hi, const int *map_dev,
                      const double *const *fl_hi, double *const *alm_out, const double *const *q, int lmin, double *pre_dev, void *stream);
int pl_alm_splice_dot_b(int lmax_lo, int nb, const double *alm_lo, int lmax_hi, const double *alm_hi, const double *fl_hi, int lsplit, double *out,
                        const double *q, int lmin, double *pre_dev, void *stream);

/* Copy of ndoubles doubles from device memory to `dst` -- device memory or pinned (device-mapped) host memory -- by a kernel of
 * `nblocks` workgroups on `stream` (16-byte aligned pointers).  Used for the estimator outputs (the reference writes them
 * to disk, qest.py:325-331): a few workgroups saturate PCIe with posted writes and leave the compute units to the kernels of the
 * next reconstruction, which the runtime's own blit copy does not. */
int pl_copy_slim(const double *src_dev, double *dst, int64_t ndoubles, int nblocks, void *stream);

/* Pixel-space helpers (qest.py:256-257,276-278; opfilt_tt.py:195, opfilt_pp.py:276-299). */
/* out = a * b (element-wise, n doubles) */
int pl_map_mul(int64_t n, const double *a, const double *b, double *out, void *stream);
/* (Q, U) <- (nqq Q + nqu U, nqu Q + nuu U) in place: the polarization inverse-noise weighting with a QU cross term
 * (alm_filter_ninv.apply_map, opfilt_pp.py:295-300, opfilt_tp.py:321-326) in one pass over the pixels */
int pl_map_qu_weight(int64_t n, double *qmap, double *umap, const double *nqq, const double *nqu, const double *nuu, void *stream);
/* complex product of spin maps: (or + i oi) (+)= sign * (ar + s1 i ai)(br + s2 i bi); accumulate != 0 adds into out */
int pl_map_cmul(int64_t n, const double *ar, const double *ai, double s1, const double *br, const double *bi, double s2,
                double sign, double *outr, double *outi, int accumulate, void *stream);

/* Real-space product of the lensing quadratic estimators in one pass over the pixels (qest.py:254-257 temperature part,
 * :273-278 polarization part, :318-322 their sum for the minimum-variance estimator):
 *   out_re + i out_im = (rep - i imp)(g3 + i c3) - (rep + i imp)(g1 - i c1) + tmap (gt + i ct)
 * rep, imp: spin-2 inverse-variance filtered map; (g3, c3), (g1, c1): spin-3 and spin-1 gradient legs; tmap: filtered
 * temperature; (gt, ct): spin-1 temperature gradient leg.  Pass tmap = NULL or rep = NULL to leave a part out. */
int pl_qe_lens_product(int64_t n, const double *tmap, const double *gt, const double *ct, const double *rep, const double *imp,
                       const double *g3, const double *c3, const double *g1, const double *c1, double *out_re, double *out_im,
                       void *stream);

/* Simulation inputs generated on the device (replaces numpy's standard_normal in plancklens/sims/phas.py:125-195 and
 * the noise adds of sims/maps.py:46-77,136-173).  Standard normal deviates are a pure function of (key, position): Philox4x32-10 on the
 * counter (position / 2, tag), 2 x 53-bit uniforms, Box-Muller; position 2 p takes the cosine deviate, 2 p + 1 the sine one.
 *   pl_map_add_normal : map_out[i] = (map_in ? map_in[i] : 0) + sigma n_i, i < n (tag 0); map_out may be map_in; device pointers,
 *                       16-byte aligned.  One pass over the map, no array of deviates.
 *   pl_alm_unit_phases: unit-variance alm (healpy layout, mmax = lmax) under `key` (tag 1): entries of m > 0 are (n_2i, n_2i+1) / sqrt 2,
 *                       the m = 0 column is real with unit variance (the convention of phas.py:162-168). */
int pl_map_add_normal(int64_t n, const double *map_in, double *map_out, double sigma, uint64_t key, void *stream);
int pl_alm_unit_phases(int lmax, double *alm_out, uint64_t key, void *stream);

/* FP64 FMA-rate microbenchmarks (16 independent chains per lane, no memory traffic): achieved TFLOP/s.
 * pl_fma64_rate_tflops: mode 0 = one vector + one scalar source besides the accumulator, 1 = two scalar (wave-uniform)
 * sources -- the operand mix of the synthesis kernels --, 2 = three vector sources -- the analysis kernels.
 * (The probes of the FP64 matrix pipe live outside the library: tools/probes/mfma64_probe.hip.)
 * pl_fma64_peak_tflops = mode 1, the highest of the vector modes. */
double pl_fma64_rate_tflops(int mode, int iters, void *stream);
double pl_fma64_peak_tflops(int iters, void *stream);

#ifdef __cplusplus
}
#endif
#endif
