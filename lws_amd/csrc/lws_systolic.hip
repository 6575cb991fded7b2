// lws_systolic.hip -- the fast path for batch LWS (LWSQ2 / LWSQ4 / LWSanyQ, lwslib.cpp:72-373) on
// gfx950: an order-exact *systolic* re-statement of the in-place Gauss-Seidel sweep.
//
// One workgroup (8 waves = 2 per SIMD: 7 sweep slots and a service wave that loads from HBM, computes the Nyquist
// bins and writes the results back) owns one spectrogram and keeps I = 7 consecutive sweeps in flight.  A lane is a
// (sweep, frame) processor that marches along the bins of its frame,
// one bin per step (two per rendez-vous: pairs of an even and an odd bin); the 64 lanes of compute wave i work on 64 consecutive frames of sweep
// "iteration g*I + i", frame m trailing frame m-1 by SKEW = 8 bins, and sweep j+1 trailing sweep j
// by LAG = 32 steps:
//
//      bin (frame m, bin c) of the sweep handled by wave i runs at step  t = 8*m + c + 32*(i+1) (+ group offset)
//
// SKEW > L and LAG > L + (Q-1)*SKEW guarantee that every "new" neighbour -- (m, c-k), (m-r, c+-k) --
// has already been produced and every "old" neighbour -- (m, c+k), (m+r, c+-k) -- has not yet been
// overwritten by this sweep but has been finished by the previous one: exactly the values the
// sequential reference reads (same argument as the skewed wavefront of lws_generic.hip; SURVEY.md
// facts 1 and 12).  A produced value is consumed by its 7 consumer lanes within 31 steps, so the
// whole exchange runs through per-wave rings in LDS:
//
//      ring[set][step mod 32][lane]  (float2)      set s = output of compute wave s-1, set 0 = loader
//
// i.e. a value is addressed by WHEN it was produced, not by where it lives in the spectrogram, and a
// reader's address is  lane_base + ((t + const) mod 32)*512  with a compile-time `const` per stencil
// tap (the step loop is unrolled by 8 so that the bin phase -- and with it bin % Q, the twiddle of
// the weights and the ring slot -- is static).  HBM is touched once per I sweeps: the service wave
// streams the spectrogram in (set 0) ahead of wave 0, the last wave streams it out, both through a
// time-skewed global layout  state_w[(8m + c) mod G][m mod 64]  in which every access of a wave is
// 64 consecutive elements.
//
// Frequency edges: the Hermitian images below DC / above Nyquist live in two pseudo-lanes of every ring row, written
// (conjugated) by the lane that produces the mirrored bin; the one lane of a wave that is within L bins of a frame
// edge reads its image taps there.  The Nyquist bin (bin F-1) does not fit the 512-step frame period and is
// computed by the service wave (one lane per sweep in flight), which also runs the loader.
//
// Scope of this file (one source, fifteen builds: the -D switches below): weights with the twiddle structure create_weights produces
// (lws.pyx:160-181: W[p][r][k] = W[0][r][k] exp(2j pi p r s / P), summarised or general tensors), fp32 arithmetic, fp32 or fp16
// storage, F-1 even (a multiple of 8, or a frame end inside a block of 8 steps: one instantiation per phase, th0) and >= 16:
//   static twiddles (P = Q, s = 1): Q in {2,4}, L <= 5, F-1 <= 512 (narrow; half / quarter: <= 256 / 128 with 2 / 4 sweep slots
//     per wave; wide / xwide: <= 1024 / 2048 with 2 / 4 waves per slot; l7: L in {6,7}; r16 variants: Q = 2 on a 16-step ring);
//     Q = 8, L <= 5, F-1 <= 512 (q8: 64-step ring, a main and two helper waves per sweep slot);
//   table twiddles: tw, tw_half, tw_wide -- Q in {3,4} with any P <= 128: Q = 3, and the general weights of a hop that does not
//     divide the frame -- L <= 5, F-1 <= 1024; tw_q8 -- 5 to 8 frames per stencil row (Q in {5,6,7}, fractional Q above 4) on the
//     Q = 8 build's geometry, L <= 5, F-1 <= 512.
// Anything else (hop < frame / 8, L >= 8, F-1 > 2048, weights without the structure, fp64) is served by the generic engine.
#include "lws_common.h"
#include "lws_systolic.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

// The file is compiled twice: as is (frames of up to 513 bins: a round of 64 frames, one wave per sweep slot, 7 sweep slots)
// and with -DLWS_WIDE=1 into namespace lws::wide (frames of up to 1025 bins: a lane needs 1024 steps per frame, so a round
// is 128 frames = TWO waves per sweep slot on a ring row of 128 lanes, with the same 8-step skew, 32-step lag and ring
// depth; 3 sweep slots (6 compute waves) and two service waves, one per half of the row, fit the LDS).
#ifndef LWS_WIDE
#define LWS_WIDE 0
#endif
// ... and a third time with -DLWS_Q8=1 into namespace lws::q8 for Q = 8 (hop = window / 8, frames of up to 513 bins): the
// taps reach 7 frames either way, so the lag between sweeps and the ring are 64 steps deep (LAG > 8 (Q-1) + L), the halo is
// 7 lanes, and 2 sweep slots (one wave each) fit the LDS.  Half of the twiddles exp(2j pi (bin % 8) r / 8) are odd eighth
// turns: a second weight set, W[0][r][k] exp(j pi / 4), serves those, and the 96 weights live in VGPRs (three waves per
// workgroup, one per SIMD: registers are plentiful, SGPRs are not).
#ifndef LWS_Q8
#define LWS_Q8 0
#endif
#ifndef LWS_SPLIT_ROLES
#define LWS_SPLIT_ROLES 1
#endif
// ... and with -DLWS_SPW=2 (namespace lws::half) / -DLWS_SPW=4 (lws::quarter) for SHORT frames, of up to 257 / 129 bins (512- and
// 256-point STFTs: 16 kHz speech).  A lane of the narrow build needs 512 steps per frame whatever its length, so a 257-bin
// frame left it idle half of the time.  Here a round is 32 (16) frames = 256 (128) steps, and a wave carries SPW sweep slots
// side by side: lanes [32 s, 32 s + 32) of compute wave w are the 32 frames of sweep slot w SPW + s, which trails slot
// w SPW + s - 1 -- its neighbours in the same wave, or the last slot of the wave before -- by the usual 32 steps.  Same skew,
// lag, ring depth and per-bin code; 14 (24) sweep slots per pass over HBM instead of 7; the two halves of a wave are always
// in the same pair, so the flow control between waves is unchanged.
#ifndef LWS_SPW
#define LWS_SPW 1
#endif
// ... and with -DLWS_L7=1 (namespace lws::l7) for stencils of half-width 6 and 7 (Q in {2, 4}, frames of up to 513 bins).  A lane
// works on two bins per rendez-vous, so the newest tap of a pair's second bin, (m-1, c+1+L), must be two steps old when the
// pair starts: with the usual skew of 8 steps between frames that holds for L <= 5 only.  Here frame m trails frame m-1 by
// SKEW = 16 steps (two blocks): the tap is 8 steps old, no frame is "late", the lag between sweeps and the ring are 64 steps
// (LAG > 16 (Q-1) + L), three sweep slots fit the LDS, and a lane's frame period is 1024 steps -- a lane is idle at least
// half of the time.  A quarter of the narrow build's rate; the generic engine these shapes used to get is 5x slower still.
#ifndef LWS_L7
#define LWS_L7 0
#endif
// ... and with -DLWS_R16=1 (namespace lws::q2) for Q = 2 (hop = half a frame, the reference's LWSQ2; frames of up to 513 bins).  The
// taps reach one frame either way, so a lag of 16 steps between sweeps is enough (LAG > 8 (Q-1) + L = 13) and the ring is 16
// steps deep: sets of 9 KB instead of 18, FIFTEEN sweep slots in the same LDS -- one wave each, four waves per SIMD (the Q = 2
// kernels need ~115 VGPRs).  Throughput follows the number of waves that run a dependent chain.
#ifndef LWS_R16
#define LWS_R16 0
#endif
// (LWS_R16 also combines with LWS_WIDE=1 -- namespace lws::wide_q2, frames of up to 1025 bins: seven sweep slots of two waves
//  instead of three)
// (... and with LWS_SPW=2 / 4 -- lws::half_q2, lws::quarter_q2, frames of up to 257 / 129 bins: 26 / 44 sweep slots)
// ... and with -DLWS_TW=1 (namespace lws::tw; with -DLWS_SPW=2: lws::tw_half) for weights whose twiddle is NOT a multiple of an
// eighth turn: Q = 3 (hop = a third of the frame), and the "general" weights create_weights builds for a hop that does not
// divide the frame (lws.pyx:164-181: Q' = N rows, one per bin; e.g. 25 ms frames every 10 ms, lws(400, 160): Q = 3, Qfloat = 2.5),
// the reference's LWSanyQ with Q = 3 and its LWSfractionalQ (lwslib.cpp:283-467).  Those tensors still are
//     W[p][r][k] = W[0][r][k] tau_r(p),   tau_r(p) = exp(2 pi j p r s / P),   s / P = hop / frame in lowest terms,
// so the kernel keeps the base weights W[0][r][k] in scalar registers as ever and takes tau_r(bin) -- now a per-lane value that
// no unrolling of the step loop makes static -- from a table in LDS ((P + 8) rows of Q - 1 twiddles; a lane carries the row of
// its block's first bin).  The taps of frames m-r and m+r are summed separately (U, D: every tap its own multiply-add, as
// LWSanyQ does, instead of the grouped LWSQ4 form) and enter the bin's sum as tau U + conj(tau) D.  Same schedule, rings,
// images and flow control as the narrow build; ~1.4x its instructions per tap.
#ifndef LWS_TW
#define LWS_TW 0
#endif
// (... and with LWS_Q8 -- namespace lws::tw_q8: the 64-step ring, halo of 7 and helper waves of the Q = 8 build with table twiddles, for
//  ceil(frame/hop) in 5..8 with any twiddle: Q in {5,6,7}, and fractional Q above 4.  The kernel is the Q = 8 instantiation; the frame
//  pairs the plan does not have are masked out at compile time, its pad frames and "real frame" tests follow the plan's Q, SysArgs::Qa)
// (... and with LWS_Q8 and -DLWS_TWQ=5 / 6 -- namespaces lws::tw_q5 / lws::tw_q6 (round 5): the same kernel on the ring a plan of exactly 5 / 6
//  frames per stencil row needs -- LAG > 8 (Q - 1) + L: 40 / 48 steps instead of 64 -- so that the LDS holds THREE sweep slots of a main
//  and ONE helper wave (frames m-+1 and m-+(Q-1) stay with the main wave, the helper sums the two / three pairs in between) where
//  lws::tw_q8 has two slots of a main and two helpers.  The ring is then 5 / 6 blocks of 8 steps: block indices are taken modulo NBLK
//  (blk_mod), not masked.)
#ifndef LWS_TWQ
#define LWS_TWQ 0
#endif
#if LWS_TWQ && !(LWS_TW && LWS_Q8 && (LWS_TWQ == 5 || LWS_TWQ == 6))
#error "LWS_TWQ = 5 or 6 goes with LWS_TW and LWS_Q8"
#endif
// (Measured and not kept, round 5: the same for exactly THREE frames per row -- a 24-step ring holds nine sweep slots where lws::tw has
//  seven, ten waves of <= 151 VGPRs -- lws(768,256): 8.9 ps per bin-sweep against 7.9.  A pass of nine slots takes 1.37x a pass of seven:
//  three waves on a SIMD that two already keep busy.  The shallower ring pays where sweep slots were missing, not where issue slots are.)
#if LWS_TW && (LWS_WIDE == 2 || LWS_L7 || LWS_R16 || LWS_SPW == 4 || (LWS_Q8 && (LWS_WIDE || LWS_SPW != 1)))
#error "LWS_TW goes with the narrow build, with LWS_SPW=2, with LWS_WIDE=1 or with LWS_Q8"
#endif
#if (LWS_WIDE && LWS_Q8) || ((LWS_SPW != 1 || LWS_L7) && (LWS_WIDE || LWS_Q8)) || (LWS_SPW != 1 && LWS_L7) || (LWS_R16 && (LWS_WIDE == 2 || LWS_Q8 || LWS_L7))
#error "LWS_WIDE, LWS_Q8, LWS_SPW, LWS_L7 and LWS_R16 are separate builds (LWS_R16 goes with LWS_WIDE=1 or LWS_SPW)"
#endif
#if LWS_TWQ == 5
#define LWS_NS_OPEN namespace lws { namespace tw_q5 {
#define LWS_NS_CLOSE } }
#elif LWS_TWQ == 6
#define LWS_NS_OPEN namespace lws { namespace tw_q6 {
#define LWS_NS_CLOSE } }
#elif LWS_TW && LWS_Q8
#define LWS_NS_OPEN namespace lws { namespace tw_q8 {
#define LWS_NS_CLOSE } }
#elif LWS_TW && LWS_SPW == 2
#define LWS_NS_OPEN namespace lws { namespace tw_half {
#define LWS_NS_CLOSE } }
#elif LWS_TW && LWS_WIDE
#define LWS_NS_OPEN namespace lws { namespace tw_wide {
#define LWS_NS_CLOSE } }
#elif LWS_TW
#define LWS_NS_OPEN namespace lws { namespace tw {
#define LWS_NS_CLOSE } }
#elif LWS_R16 && LWS_WIDE
#define LWS_NS_OPEN namespace lws { namespace wide_q2 {
#define LWS_NS_CLOSE } }
#elif LWS_R16 && LWS_SPW == 2
#define LWS_NS_OPEN namespace lws { namespace half_q2 {
#define LWS_NS_CLOSE } }
#elif LWS_R16 && LWS_SPW == 4
#define LWS_NS_OPEN namespace lws { namespace quarter_q2 {
#define LWS_NS_CLOSE } }
#elif LWS_R16
#define LWS_NS_OPEN namespace lws { namespace q2 {
#define LWS_NS_CLOSE } }
#elif LWS_L7
#define LWS_NS_OPEN namespace lws { namespace l7 {
#define LWS_NS_CLOSE } }
#elif LWS_SPW == 2
#define LWS_NS_OPEN namespace lws { namespace half {
#define LWS_NS_CLOSE } }
#elif LWS_SPW == 4
#define LWS_NS_OPEN namespace lws { namespace quarter {
#define LWS_NS_CLOSE } }
#elif LWS_WIDE == 2
#define LWS_NS_OPEN namespace lws { namespace xwide {
#define LWS_NS_CLOSE } }
#elif LWS_WIDE
#define LWS_NS_OPEN namespace lws { namespace wide {
#define LWS_NS_CLOSE } }
#elif LWS_Q8
#define LWS_NS_OPEN namespace lws { namespace q8 {
#define LWS_NS_CLOSE } }
#else
#define LWS_NS_OPEN namespace lws {
#define LWS_NS_CLOSE }
#endif

LWS_NS_OPEN
namespace {

constexpr int LANES = 64;                                // lanes of a wave
// (-DLWS_WIDE=2, namespace lws::xwide: frames of up to 2049 bins -- a 4096-point STFT -- on a ring row of 256 lanes = FOUR waves per
//  sweep slot; the LDS holds the loader's set and one slot's, so every sweep is a pass over HBM, with four compute waves each
//  alone on its SIMD beside a service wave: a quarter of the narrow build's rate per bin, four times the generic engine's)
constexpr int WPS = LWS_WIDE == 2 ? 4 : (LWS_WIDE ? 2 : 1);   // waves per sweep slot
constexpr int SPW = LWS_SPW;                             // sweep slots per wave (short-frame builds)
constexpr int ROWL = LANES * WPS / SPW;                  // lanes (frames) of a ring row = frames of a round
constexpr int ROWL_SHIFT = LWS_WIDE == 2 ? 8 : (LWS_WIDE ? 7 : (SPW == 4 ? 4 : (SPW == 2 ? 5 : 6)));
static_assert((1 << ROWL_SHIFT) == ROWL, "row length");
constexpr int RING = LWS_TWQ ? 8 * LWS_TWQ : ((LWS_Q8 || LWS_L7) ? 64 : (LWS_R16 ? 16 : 32));
constexpr int NBLK = RING / 8;                           // ring blocks of 8 steps
// block index of the ring: x mod NBLK (a mask when NBLK is a power of two; lws::tw_q5 / tw_q6 have 5 / 6 blocks)
__host__ __device__ constexpr int blk_mod(int x) { return (NBLK & (NBLK - 1)) == 0 ? (x & (NBLK - 1)) : ((x % NBLK) + NBLK) % NBLK; }
// ring entry of production time nu, lane l:  set + ((nu >> 1) & 15) * PAIR_BYTES + (l + HALO) * 16 + (nu & 1) * 8
// -- two consecutive times of one lane share a 16-byte cell, so a reader fetches two adjacent taps with one
// ds_read_b128; every row of 64 lanes carries HALO copies of the opposite end on each side (lanes -3..-1 mirror
// 61..63, lanes 64..66 mirror 0..2), so "the lane d frames above / below" is a compile-time address offset and a
// lane needs one base register per ring block instead of one per (neighbour, block).  Two more pseudo-lanes per row
// hold the Hermitian images: lane PLL the bins -1..-L of the frames (at the production times those bins would have,
// 8m - j), lane PLR the Nyquist bin and the bins above it (times 8m + C + j), written conjugated by the lane that
// produces the mirrored bin -- the reference's pad columns (lwslib.cpp:362-367), kept in time coordinates.  A lane
// near a frame edge reads those cells instead of its neighbour lane's: same compile-time offsets, other base.
constexpr int SLOT_BYTES = ROWL * 8;                     // Nyquist buffer: bytes per set (one float2 per row lane)
constexpr int HALO = LWS_Q8 ? 7 : 3;                     // >= Q - 1
constexpr int NDR = 2 * HALO + 1;                        // neighbour frames -HALO .. HALO
constexpr int QMAX = HALO + 1;
constexpr int LANE_B = 16;
constexpr int PLL = ROWL + 2 * HALO, PLR = PLL + 1;       // absolute row indices of the two image pseudo-lanes
constexpr int PAIR_BYTES = (ROWL + 2 * HALO + 2) * LANE_B;   // two consecutive times x 72 (136) row entries
constexpr int BLK_BYTES = 4 * PAIR_BYTES;                // one block of 8 steps
constexpr int SET_BYTES = (RING / 2) * PAIR_BYTES;       // 18 KiB (wide: 34 KiB, Q = 8: 40 KiB)
#ifndef LWS_NSLOTS
#define LWS_NSLOTS ((LWS_R16 && LWS_WIDE) ? 7 : (LWS_R16 && LWS_SPW == 2) ? 26 : (LWS_R16 && LWS_SPW == 4) ? 44 : LWS_R16 ? 15 : LWS_L7 ? 3 : LWS_WIDE == 2 ? 1 : LWS_WIDE ? 3 : (LWS_TWQ ? 3 : LWS_Q8 ? 2 : (LWS_SPW == 4 ? 24 : 7 * LWS_SPW)))   // (SPW = 4: 25 ring sets of 6 KB are what the LDS holds -- six compute waves)
#endif
constexpr int NSLOTS = LWS_NSLOTS;                       // sweeps in flight (compute waves)
constexpr int NSETS = NSLOTS + 1;
constexpr int NYQ_OFF = NSETS * SET_BYTES;               // Nyquist values: [set][lane] float2
constexpr int THR_OFF = NYQ_OFF + NSETS * SLOT_BYTES;    // effective thresholds: floats
constexpr int MAX_ITERS = 440;
constexpr int META_OFF = THR_OFF + MAX_ITERS * 4;        // n_eff
constexpr int DONE_OFF = META_OFF + 16;               // per-wave count of completed steps (flow control)
constexpr int DUMMY_OFF = DONE_OFF + 64;               // 64 x 8 B: where predicated-off lanes park their conditional writes
constexpr int SCRATCH_OFF = DUMMY_OFF + LANES * 8;     // where the compute lanes that have no image to publish store instead (see image_base)
constexpr int SCRATCH_BYTES = 4 * PAIR_BYTES + LANES * 8;
// Q = 8: a sweep slot is one MAIN wave (centre frame, frames m-+1 and m-+7, re-projection, publishing) and NHELP helper
// waves that sum the taps of the other neighbour frames 4 or 6 steps ahead of it (help_ahead) -- every tap of frames m-+2 .. m-+6 is
// at least 10 steps old when its bin is due -- and leave the two sums of a pair of bins in a mailbox (4 pairs deep).  The
// sums of a bin are the bulk of its ~310 instructions and one wave per SIMD issues one every ~4.4 clocks: spreading a slot
// over three waves is what fills the other SIMDs (2 slots are all the LDS holds at this ring depth).
constexpr int NHELP = LWS_TWQ ? 1 : (LWS_Q8 ? 2 : 0);    // helper waves per sweep slot
// how far ahead helper h works: two pairs.  (Three -- 6 steps, so that a helper sums in the pair in which the main wave of its
// SIMD has little to do -- is legal but slower: the waves of a slot meet at every pair, and then every pair is a heavy one
// for somebody: 143 -> 165 ms.)
__host__ __device__ constexpr int help_ahead(int h) { return h >= 1 ? 4 : 4; }
constexpr int MBOX_OFF = SCRATCH_OFF + SCRATCH_BYTES;    // [slot][helper][pair & 3][lane]: (sum of the first bin, of the second)
constexpr int MBOX_BYTES = NSLOTS * NHELP * 4 * LANES * 16;
constexpr int WNYQ_OFF = MBOX_OFF + MBOX_BYTES;          // Q = 8: the Nyquist lanes' weights (the waves keep only their own in registers)
constexpr int TWNYQ_OFF = WNYQ_OFF + (LWS_Q8 ? QMAX * 6 * 8 : 0);   // Q = 8 with table twiddles: tau_r(F-1), r = 0..7, for the Nyquist lanes
                                                                     // (a per-lane index: from LDS, like their weights)
constexpr int LDS_BASE_BYTES = TWNYQ_OFF + ((LWS_Q8 && LWS_TW) ? 8 * 8 : 0);
// LWS_TW: the twiddle table, [row][r - 1] float2 with row = bin mod P, rows 0 .. P + 7 (a lane holds the row of its block's first
// bin; the other seven bins of the block are compile-time offsets from it); a row is TW_ROW bytes: tau_1, tau_2, tau_3, unused
constexpr bool TW = LWS_TW != 0;
constexpr int TW_ROW = LWS_Q8 ? 64 : 32;              // (the 64-step-ring build: tau_1 .. tau_7, unused)
constexpr int TW_OFF = (LDS_BASE_BYTES + 15) & ~15;
constexpr int TW_PMAX = TW ? ((160 * 1024 - TW_OFF) / TW_ROW - 8 < 128 ? (160 * 1024 - TW_OFF) / TW_ROW - 8 : 128) : 0;   // longest twiddle period served
constexpr int LDS_BYTES = TW ? TW_OFF + (TW_PMAX + 8) * TW_ROW : LDS_BASE_BYTES;
static_assert(!TW || TW_PMAX >= 16, "no room for a twiddle table");
__host__ __device__ constexpr int mbox_addr(int slot, int h, int pair) { return MBOX_OFF + ((slot * NHELP + (h - 1)) * 4 + (pair & 3)) * LANES * 16; }
constexpr int SKEW = LWS_L7 ? 16 : 8, ROWP = SKEW * ROWL, LAG = RING;
constexpr int LATE_DN = LAG / SKEW - 1;                  // frame m + LATE_DN of the previous sweep is only SKEW steps ahead of a lane (as frame m - 1 of its own sweep is)
// weights a kernel carries.  Q = 8: every wave of a slot keeps only the weights of the frames it sums (row_owner), both weight
// sets for the odd frames: 3 lists of up to WLIST entries, then the 48 weights of set 0 in full for the Nyquist lanes
constexpr int WLIST = 32, WNYQ = 3 * WLIST;
constexpr int NW = LWS_Q8 ? WNYQ + QMAX * 6 : 4 * 8;
__host__ __device__ constexpr int row_owner(int R) {     // which wave of a slot sums frames m-+R: 0 = main, h = helper h
    return (NHELP == 0 || R <= 1 || R == LATE_DN) ? 0 : (NHELP == 1 ? 1 : (R <= 4 ? 1 : 2));
}
constexpr int ROWP_SHIFT = ROWL_SHIFT + (LWS_L7 ? 4 : 3);
static_assert((1 << ROWP_SHIFT) == ROWP, "frame period");
static_assert(NSLOTS % SPW == 0 && (SPW == 1 || (WPS == 1 && NHELP == 0)) && ROWL >= 2 * HALO + 2, "slots per wave");
constexpr int NCOMPUTE = NSLOTS * WPS / SPW;             // compute waves; roles NCOMPUTE .. NCOMPUTE + WPS - 1 are the service waves
constexpr int NHELPERS = NSLOTS * NHELP;                 // roles NCOMPUTE + WPS .. : helper h of slot s is role NCOMPUTE + WPS + s * NHELP + h - 1
constexpr int NWAVES = NCOMPUTE + WPS + NHELPERS;
constexpr int NTHREADS = LANES * NWAVES;
static_assert(NWAVES <= 16, "one progress counter per wave");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

template <int... Is, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F &&>(f));
}

// Where a stencil tap is found.  off: production time relative to the reader's clock (already
// includes -LAG for "old" values); set_new: ring set of the reader's own sweep (1) or of the
// previous sweep (0).
enum { K_RING = 0, K_NYQ = 1, K_SELF = 2, K_NEXT = 3, K_PREV = 4 };
// register sources: K_SELF / K_NEXT = previous-sweep value of the own bin c / c+1 (prefetched), K_PREV = this sweep's
// value of bin c-1 (the lane's own previous output; for the second bin of a pair it is not in LDS yet)
struct Src { int kind, set_new, off, conj; };

__host__ __device__ constexpr Src src_normal(int dr, int dk) {
    if (dr == 0 && dk == 1) return Src{K_NEXT, 0, 0, 0};
    if (dr == 0 && dk == -1) return Src{K_PREV, 0, 0, 0};
    const bool is_new = dr < 0 || (dr == 0 && dk < 0);
    return Src{K_RING, is_new ? 1 : 0, SKEW * dr + dk - (is_new ? 0 : LAG), 0};
}
// reader at bin c = P (first bins of a frame), tap at c + dk < 0: image of bin -(c+dk), conjugated
__host__ __device__ constexpr Src src_start(int P, int dr, int dk) {
    const int cm = -(P + dk);            // mirrored bin, 1..L
    const int off0 = SKEW * dr + cm - P;
    if (dr == 0 && cm == P) return Src{K_SELF, 0, 0, 1};
    if (dr == 0 && cm == P + 1) return Src{K_NEXT, 0, 0, 1};
    if (dr == 0 && cm == P - 1) return Src{K_PREV, 0, 0, 1};
    const bool is_new = dr < 0 || (dr == 0 && cm < P);
    return Src{K_RING, is_new ? 1 : 0, off0 - (is_new ? 0 : LAG), 1};
}
// reader at bin c = C - e (last bins of a frame), tap at c + dk >= C: the Nyquist bin (dk == e) or
// the image of bin 2C - (c+dk), conjugated
__host__ __device__ constexpr Src src_end(int e, int dr, int dk) {
    if (dk == e) return Src{K_NYQ, dr < 0 ? 1 : 0, 0, 0};
    const int d = 2 * e - dk;            // mirrored bin minus reader bin
    if (dr == 0 && d == 0) return Src{K_SELF, 0, 0, 1};
    if (dr == 0 && d == 1) return Src{K_NEXT, 0, 0, 1};
    if (dr == 0 && d == -1) return Src{K_PREV, 0, 0, 1};
    const bool is_new = dr < 0 || (dr == 0 && d < 0);
    return Src{K_RING, is_new ? 1 : 0, SKEW * dr + d - (is_new ? 0 : LAG), 1};
}

// Frame ends.  RE = (F-1) mod 8, even.  The block of a frame that holds its last bin C-1 -- the "end block" -- sees bin C at
// block-relative position th0 = 8 (RE = 0: the frame ends with a full block) or RE (the last block has only RE bins, its
// phases RE..7 are dead: they run like any other step, on a target magnitude of -inf, and what they leave in the ring is
// never read); the block before it sees bin C at th1 = th0 + 8, which matters when RE < L.
__host__ __device__ constexpr int th0(int RE) { return RE ? RE : 8; }
__host__ __device__ constexpr int th1(int RE) { return th0(RE) + 8; }
// which images the bin at phase PH leaves behind: bin -PH (first block of a frame), bin C + j0 (end block), bin C + j1 (the
// block before it)
template <int L, int PH, int RE> struct ImagePhase {
    static constexpr bool lo = (PH >= 1 && PH <= L);
    static constexpr int j0 = th0(RE) - PH, j1 = th1(RE) - PH;
    // (RE != 0: bin C+L is a tap of nobody below bin C -- the Nyquist lanes take their taps from the bins themselves -- and is
    // not kept; the RE = 0 build stores it as it always did)
    static constexpr int JMAX = RE ? L - 1 : L;
    static constexpr bool hi0 = (j0 >= 1 && j0 <= JMAX), hi1 = (RE != 0 && j1 >= 1 && j1 <= JMAX);
};

struct SysArgs {
    // storage format of the skewed layout (template parameter H16 of the kernels): fp32 -- float2 / float -- or
    // fp16 -- half2 / half, scaled per spectrogram by store_scale(amax) (LWS_STORAGE_FP16)
    void *state_w;           // [B][G][64]   time-skewed spectrogram
    const void *amp_w;       // [B][G][64]
    void *state_nyq;         // [B][TpPad]
    const void *amp_nyq;     // [B][TpPad]
    const float *thr;        // [B][n_iters] thresholds scaled by mean|S|
    const float *amax;       // [B] max target magnitude
    int n_iters, T, Tp, TpPad, Kr, G, C;
    int nwg;                 // workgroups per spectrogram (passes over HBM are dealt round-robin to them)
    unsigned *progress;      // [B][nwg] rows of the skewed state each workgroup has completed (nwg > 1)
    int *err;                // set if a workgroup gave up waiting for its producer
    const int *gate;         // non-null: run only if *gate != 0 (the single-workgroup re-run after a hand-over time-out)
    int spin_limit;          // polls of a producer's counter before a workgroup gives up
    int rolemap;             // experiment hook (LWS_SYSTOLIC_ROLEMAP): which hardware wave takes which role, wide build
    int stress;              // test hook (LWS_SYSTOLIC_STRESS): role mask | pair << 16 -- the waves of the mask stall ~10 us
                             // before that pair of every block; the flow control must make the results independent of it
    // LWS_TW: twiddles tau_r(p) = exp(2 pi j p r s / P) of general weights
    const float *tw_table;   // device: [(P + 8)][4] float2, row p: tau_1(p), tau_2(p), tau_3(p), 0
    int tw_P;                // period of the twiddles in bins
    float tw_invP;
    unsigned long long tw_nyq[8];  // tau_r(F-1), r = 0..7, for the Nyquist lanes: bit patterns of (re, im)
    int Qa;                  // the plan's Q (frames of a stencil row): the template Q, except in the build lws::tw_q8, whose kernel is the
                             // Q = 8 instantiation for every Q in 5..8
    unsigned long long w[NW];      // W[0][r][k], r < Q, k <= L (at most 4 x 8): bit patterns of (re, im) as one 64-bit scalar;
                                   // Q = 8: [set][r][k] with set 1 = W[0][r][k] exp(j pi / 4), see widx()
};

// volatile: keeps every tap a separate ds_read_b64 (the backend otherwise fuses pairs into
// ds_read2st64_b64, which moves half the bytes per LDS cycle -- MI355X_MICROARCH.md, LDS table)
__device__ __forceinline__ float2 lds_read(int addr) {
    // `addr` is a byte offset into the dynamic LDS segment, which starts at LDS address 0 (the kernel
    // has no static __shared__ objects); address space 3 keeps it a ds_ instruction
    using lds_u64 = const volatile __attribute__((address_space(3))) unsigned long long;
    const unsigned long long u = *(lds_u64 *)(unsigned)addr;
    return make_float2(__uint_as_float((unsigned)(u & 0xffffffffull)), __uint_as_float((unsigned)(u >> 32)));
}
__device__ __forceinline__ void lds_write(int addr, float2 v) {
    // volatile, like the reads: program order of all ring traffic is what the flow control below relies on
    using lds_u64w = volatile __attribute__((address_space(3))) unsigned long long;
    *(lds_u64w *)(unsigned)addr = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
}
__device__ __forceinline__ int lds_read_i32(int addr) {
    using lds_i32 = const volatile __attribute__((address_space(3))) int;
    return *(lds_i32 *)(unsigned)addr;
}
__device__ __forceinline__ void lds_write_i32(int addr, int v) {
    using lds_i32w = volatile __attribute__((address_space(3))) int;
    *(lds_i32w *)(unsigned)addr = v;
}
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f lds_read128(int addr) {
    using lds_v4 = const volatile __attribute__((address_space(3))) v4f;
    return *(lds_v4 *)(unsigned)addr;
}
__device__ __forceinline__ float2 cj(float2 v) { return make_float2(v.x, -v.y); }

// ---- storage formats of the skewed HBM layout --------------------------------------------------------------------
// H16 = false: float2 state, float magnitudes (20 B per active bin and sweep).  H16 = true (LWS_STORAGE_FP16): half2 state
// and half magnitudes (10 B), both multiplied by store_scale(largest magnitude of the spectrogram), a power of two that
// brings the data to [0, 2): exact, and the whole kernel then works in that scaled domain (its thresholds are scaled the
// same way; the weighted sums are only ever normalised, so nothing else changes).  Arithmetic and the LDS rings are fp32
// in both formats; the rounding to half happens once per pass over HBM (NSLOTS sweeps).
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
__host__ __device__ __forceinline__ float store_scale(float amax) {
    unsigned b;
    __builtin_memcpy(&b, &amax, 4);
    const unsigned e = (b >> 23) & 0xffu;
    const unsigned sb = (e == 0u || e == 0xffu) ? 0x3f800000u : ((e >= 254u ? 1u : 254u - e) << 23);
    float sc;
    __builtin_memcpy(&sc, &sb, 4);
    return sc;
}
__device__ __forceinline__ float2 unpack_h2(unsigned u) {
    const h2_t h = __builtin_bit_cast(h2_t, u);
    return make_float2((float)h.x, (float)h.y);
}
__device__ __forceinline__ unsigned pack_h2(float2 v) {   // round to nearest even
    const h2_t h = {(_Float16)v.x, (_Float16)v.y};
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float unpack_h(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }
__device__ __forceinline__ unsigned short pack_h(float v) { return __builtin_bit_cast(unsigned short, (_Float16)v); }
template <bool H16> struct Store {
    using cplx = float2; using real = float;
    static constexpr int CB = 8, RB = 4;
};
template <> struct Store<true> {
    using cplx = unsigned; using real = unsigned short;
    static constexpr int CB = 4, RB = 2;
};

// A complex value as it travels from the global load to its use one block later: the raw bits (a conversion at the load
// would wait for it there and then).
template <bool H16> __device__ __forceinline__ float2 raw_value(float2 raw) { return H16 ? unpack_h2(__float_as_uint(raw.x)) : raw; }
template <bool H16> __device__ __forceinline__ float raw_real(float raw) { return H16 ? unpack_h((unsigned short)__float_as_uint(raw)) : raw; }

// Loads that must observe what another wave of this workgroup stored earlier: bypass the per-CU L1.  Returns raw bits.
template <bool H16> __device__ __forceinline__ float2 load_l2(const void *base, size_t idx) {
    if constexpr (H16) {
        const unsigned u = __hip_atomic_load(static_cast<const unsigned *>(base) + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_float2(__uint_as_float(u), 0.f);
    } else {
        const unsigned long long u =
            __hip_atomic_load(static_cast<const unsigned long long *>(base) + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float2 v;
        v.x = __uint_as_float((unsigned)(u & 0xffffffffull));
        v.y = __uint_as_float((unsigned)(u >> 32));
        return v;
    }
}
template <bool H16> __device__ __forceinline__ float load_real_raw(const void *base, size_t idx) {
    if constexpr (H16) return __uint_as_float((unsigned)static_cast<const unsigned short *>(base)[idx]);
    else return static_cast<const float *>(base)[idx];
}

// Stores another workgroup (possibly on another XCD, behind another L2) will read during this launch: write through.
// (With one workgroup per spectrogram producer and consumer share the CU's XCD and a plain store is enough.)
template <bool H16> __device__ __forceinline__ void store_l2(void *base, size_t idx, float2 v, bool shared) {
    if constexpr (H16) {
        const unsigned u = pack_h2(v);
        if (shared) __hip_atomic_store(static_cast<unsigned *>(base) + idx, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else static_cast<unsigned *>(base)[idx] = u;
    } else if (shared) {
        const unsigned long long u = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
        __hip_atomic_store(static_cast<unsigned long long *>(base) + idx, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        static_cast<float2 *>(base)[idx] = v;
    }
}

// ---- synchronisation between the waves of a workgroup -----------------------------------------------------
// The step loop works on PAIRS of consecutive bins (phases 0+1, 2+3, 4+5, 6+7): every tap of both bins, except the
// lane's own previous output, was produced before the pair starts (the newest one -- frame m-1, tap +L of the second bin --
// in the pair before), and nothing older than RING - 2 steps is ever read while a ring slot is rewritten after RING.  A wave may therefore start a pair as soon as the waves it
// exchanges data with -- its producer (previous slot, or the service wave), its consumer (next slot) and the
// service wave -- have completed the previous pair.  Each wave publishes the first step it has not completed yet
// in LDS; LDS executes the operations of one wave in program order and all ring accesses are volatile (compiler
// order), so "write data, then the counter" / "read the counter, then the data" is sufficient.  No barriers.
__device__ __forceinline__ void flow_wait(int lane, int s, bool watched) {
    const int addr = DONE_OFF + (lane & 15) * 4;   // lane l < number of waves watches wave l
    while (true) {
        const int v = lds_read_i32(addr);
        if (__all(!watched || v >= s)) break;
    }
}
__device__ __forceinline__ void flow_publish(int lane, int wave, int s_next) {
    if (lane == 0) lds_write_i32(DONE_OFF + wave * 4, s_next);
}

// Per-lane registers of a compute lane that stay valid for one block of 8 steps.
struct LaneCtx {
    int obh[NBLK];    // [m]: ob[m] + halo_shift: where the halo copy of this lane's output goes (the lane's own entry if it has none)
    int ob[NBLK];     // [m]: LDS address of this lane's halo-shifted origin in the previous sweep's (old) set, block (a - m) & 3;
                      //      the own (new) set is the next one: + SET_BYTES, a compile-time offset
    int uo[NBLK];     // [m]: wave-uniform row origin (set + block) of the old set, for the image pseudo-lanes
    int nyq_base;     // NYQ_OFF + own set row + lane*8 (taps derive the neighbour lane / set from it)
    int halo_shift;   // +-ROWL lanes in bytes for the 6 lanes that also write a halo copy, else 0
    int dummy;        // private LDS slot for predicated-off conditional writes
    int lane8;
    bool is_start, is_end, live;                 // this block (8 bins of one frame); is_end: the frame's end block
    bool is_end1;                                // the block before the end block (RE != 0 builds)
    bool nxt_start, nxt_end, nxt_live;           // the following block (possibly the next frame of the lane)
    bool nxt_end1;
    float thr, nxt_thr;
    // image cells: byte offset from the lane's own origin ob[m] to the pseudo-lane's, minus what the compile-time offset
    // of the neighbour frame DR adds -- for the lane at the start (lo: PLL) / end (hi: PLR) of its frame; zero for every
    // other lane.  [DR + HALO]
    int wlo[NDR], whi[NDR];
    int whi1[NDR];                  // RE != 0: the same as whi for the lane in the block before its end block (cells from th1 on)
    int mbox;                       // Q = 8: mailbox of the slot's first helper, pair 0 (own lane)
    int tw;                         // LWS_TW: LDS address of the twiddle-table row of this block's first bin
    int tw_nxt;                     // ... and of the next block's (helper waves work ahead)
    int img_lo, img_hi, img_both;   // image_base(): row origin for the image stores of this block (phases with an image below DC / above
                                    // Nyquist / both)
};

__host__ __device__ constexpr int floor_div8(int q) { return (q >= 0) ? q / 8 : -((-q + 7) / 8); }

// write a lane's value and, for the first / last HALO lanes, its halo copy at the other end of the row (branch-free and
// without per-write address arithmetic: the halo origin obh[] is the lane's own origin shifted by +-64 lanes for those six
// lanes and unshifted for the others, so both writes use the same immediate offset)
__device__ __forceinline__ void ring_publish(int addr, int addr_halo, float2 v) {
    lds_write(addr, v);
    lds_write(addr_halo, v);   // the halo copy of the first / last HALO lanes; every other lane writes its own entry twice
}

// address of the ring entry of the lane DR frames away, produced at clock (block start + P + OFF); P may be 8
// (phase 0 of the next block).  base[m] addresses lane - HALO of block (a - m) & 3.
// NEWSET: 1 = the lane's own output set (base + SET_BYTES), 0 = the set base[] points to.
template <int P, int OFF, int DR = 0, int NEWSET = 0> __device__ __forceinline__ int ring_addr(const int (&base)[NBLK]) {
    constexpr int q = P + OFF;
    static_assert(q >= -RING && q <= 15, "ring retention exceeded");
    static_assert(DR >= -HALO && DR <= HALO, "halo too small");
    constexpr int fl = floor_div8(q);
    constexpr int m = blk_mod(-fl);        // block a+1 shares the physical block of a-(NBLK-1)
    constexpr int within = q - 8 * fl;
    return base[m] + NEWSET * SET_BYTES + (HALO + DR) * LANE_B + (within >> 1) * PAIR_BYTES + (within & 1) * 8;
}

// the same for an absolute row index (halo copies, image pseudo-lanes); base[m] is the row origin (set + block)
template <int P, int OFF, int LIDX, int NEWSET = 0> __device__ __forceinline__ int ring_addr_abs(const int (&base)[NBLK]) {
    constexpr int q = P + OFF;
    static_assert(q >= -RING && q <= (NBLK > 4 ? 31 : 15), "ring retention exceeded");   // (images are written up to 2(L-1) steps ahead of their time)
    constexpr int fl = floor_div8(q);
    constexpr int m = blk_mod(-fl);
    constexpr int within = q - 8 * fl;
    return base[m] + NEWSET * SET_BYTES + LIDX * LANE_B + (within >> 1) * PAIR_BYTES + (within & 1) * 8;
}

// Hermitian image upkeep (lwslib.cpp:362-367 in time coordinates): the lane that has just produced bin j in 1..L of
// its frame (phase PH = j, flag st) or bin C-j (phase PH = 8-j, flag en) stores the conjugate where bin -j / C+j
// would have been produced: 2j steps earlier / later.  `u` = wave-uniform row origins of the lane's output set.
// RE != 0 (see th0): the image of bin C-j goes 2j steps later whichever block bin C-j is in; en1 = the lane is in the block
// before its end block.
template <int L, int PH, int PB, int NEWSET, int RE = 0>
__device__ __forceinline__ void image_publish(const int (&u)[NBLK], bool st, bool en, int dummy, float2 out, bool en1 = false) {
    constexpr bool lo = (PH >= 1 && PH <= L), hi = (PH >= 8 - L && PH <= 7);
    if constexpr (RE != 0) {
        using IP = ImagePhase<L, PH, RE>;
        if constexpr (IP::lo || IP::hi0 || IP::hi1) {
            int addr = dummy;
            if constexpr (IP::hi1) addr = en1 ? ring_addr_abs<PB, 2 * IP::j1, PLR, NEWSET>(u) : addr;
            if constexpr (IP::hi0) addr = en ? ring_addr_abs<PB, 2 * IP::j0, PLR, NEWSET>(u) : addr;
            if constexpr (IP::lo) addr = st ? ring_addr_abs<PB, -2 * PH, PLL, NEWSET>(u) : addr;
            lds_write(addr, cj(out));
        }
    } else if constexpr (lo && hi) {   // a lane is at the start or at the end of a frame, never both
        const int a_lo = ring_addr_abs<PB, -2 * PH, PLL, NEWSET>(u), a_hi = ring_addr_abs<PB, 2 * (8 - PH), PLR, NEWSET>(u);
        lds_write(st ? a_lo : (en ? a_hi : dummy), cj(out));
    } else if constexpr (lo) {
        lds_write(st ? ring_addr_abs<PB, -2 * PH, PLL, NEWSET>(u) : dummy, cj(out));
    } else if constexpr (hi) {
        lds_write(en ? ring_addr_abs<PB, 2 * (8 - PH), PLR, NEWSET>(u) : dummy, cj(out));
    }
}

// The same store for the compute lanes without any per-store address arithmetic.  The image of the bin produced at phase PH
// goes to production time PH - 2 PH (below DC, pseudo-lane PLL) or PH + 2 (8 - PH) (above Nyquist, pseudo-lane PLR):
// 16 steps = two ring blocks apart, at the same position inside the block.  So both addresses are "row origin + the
// same compile-time offset", and the row origin -- chosen once per block and lane -- says which one it is: the block of
// the low image for a lane at the start of its frame, the block of the high image (+ one lane) for a lane at its end,
// a scratch area for everybody else.
template <int L> struct ImageBlocks {
    static constexpr int m_lo = blk_mod(-floor_div8(-1)), m_hi = blk_mod(-floor_div8(15));
    static_assert(L <= 7, "images within one block of the frame edge");
};
template <int PH, int NEWSET> __host__ __device__ constexpr int image_off() {   // offset of phase PH's image from its row origin
    constexpr int within = 8 - PH;
    return NEWSET * SET_BYTES + PLL * LANE_B + (within >> 1) * PAIR_BYTES + (within & 1) * 8;
}
template <int L, int PH, int NEWSET>
__device__ __forceinline__ void image_store(int base_lo, int base_hi, int base_both, float2 out) {
    constexpr bool lo = (PH >= 1 && PH <= L), hi = (PH >= 8 - L && PH <= 7);
    if constexpr (lo || hi) lds_write((lo && hi ? base_both : (lo ? base_lo : base_hi)) + image_off<PH, NEWSET>(), cj(out));
}

// EDGE: 0 normal, 1 first bins of a frame, 1 + e (e >= 1): last bins of a frame, the reader's bin is C - e
template <int PH, int DR, int DK, int EDGE> __host__ __device__ constexpr Src tap_src() {
    return (EDGE == 0) ? src_normal(DR, DK) : (EDGE == 1 ? src_start(PH, DR, DK) : src_end(EDGE - 1, DR, DK));
}
template <int PH, int DR, int DK, int EDGE> __host__ __device__ constexpr bool tap_in_lds() {
    constexpr Src s = tap_src<PH, DR, DK, EDGE>();
    return s.kind == K_RING || s.kind == K_NYQ;
}
// One tap fetched from LDS.  PH: bin phase the tap belongs to (decides which taps are images / the Nyquist bin);
// PB: clock of that bin relative to the start of the current block (PH, or 8 for phase 0 of the next block).
template <int PH, int PB, int DR, int DK, int EDGE>  // EDGE: see tap_src
__device__ __forceinline__ float2 tap_lds(const LaneCtx &cx) {
    constexpr Src s = tap_src<PH, DR, DK, EDGE>();
    static_assert(s.kind == K_RING || s.kind == K_NYQ, "register-sourced tap");
    float2 v;
    if constexpr (s.kind == K_NYQ) {
        // rare (one lane, last bins of a frame): Nyquist value of frame m+DR in the own / previous set
        const int ln = (cx.lane8 + 8 * DR) & (SLOT_BYTES - 1);
        v = lds_read(cx.nyq_base - cx.lane8 + ln - (s.set_new ? 0 : SLOT_BYTES));
    } else {
        static_assert(s.off <= -2 && s.off >= -(RING - 2), "tap outside ring retention (ages 2..RING-2)");
        v = lds_read(ring_addr<PB, s.off, DR, s.set_new>(cx.ob));
    }
    if constexpr (s.conj) v = cj(v);
    return v;
}
// register-sourced value of a tap (K_SELF / K_NEXT / K_PREV), conjugated if it is an image
template <int PH, int DR, int DK, int EDGE>
__device__ __forceinline__ float2 tap_reg(float2 self_old, float2 next_old, float2 prev_out) {
    constexpr Src s = tap_src<PH, DR, DK, EDGE>();
    float2 v = (s.kind == K_SELF) ? self_old : (s.kind == K_NEXT ? next_old : prev_out);
    if constexpr (s.conj) v = cj(v);
    return v;
}
template <int PH, int PB, int DR, int DK, int EDGE>
__device__ __forceinline__ float2 tap_any(const LaneCtx &cx, float2 self_old, float2 next_old, float2 prev_out) {
    if constexpr (tap_in_lds<PH, DR, DK, EDGE>()) return tap_lds<PH, PB, DR, DK, EDGE>(cx);
    else return tap_reg<PH, DR, DK, EDGE>(self_old, next_old, prev_out);
}

// Complex values travel as (re, im) register pairs (that is how ds_read_b128 delivers them), so sums and differences
// of two of them and "real scalar times complex" are single packed v_pk_add_f32 / v_pk_fma_f32 instructions (1.7x the
// float2 throughput of two scalar ones on gfx950, scratch/pk_ubench.hip).  Quarter turns, conjugation-like sign flips
// and "broadcast one half of the (re, im) weight pair" are the instructions' own operand modifiers (op_sel / neg), which
// the compiler does not derive from C++ (it builds the operands with v_mov / v_xor instead): hence the inline assembly.
// A weight is wave-uniform and stays in an aligned SGPR pair (Q = 8 build: 96 weights, in VGPRs).
#if LWS_Q8
#define LWS_WREG "v"
#else
#define LWS_WREG "s"
#endif
typedef float v2f __attribute__((ext_vector_type(2)));
using wp_t = unsigned long long;   // bit pattern of (re, im) as one 64-bit scalar
__device__ __forceinline__ v2f vv(float2 a) { return (v2f){a.x, a.y}; }
__device__ __forceinline__ float2 ff(v2f a) { return make_float2(a.x, a.y); }

// acc + (sl * w.HALF) * v'  with v' = v (SWZ 0) or (v.y, v.x) (SWZ 1) and the sign of the product chosen per lane
#define LWS_PKFMA(H, S, NL, NH)                                                                                     \
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[" #H "," #S ",0] op_sel_hi:[" #H "," LWS_NOT_##S ",1] neg_lo:[" #NL    \
        ",0,0] neg_hi:[" #NH ",0,0]"                                                                                \
        : "+v"(acc)                                                                                                 \
        : LWS_WREG(w), "v"(v))
#define LWS_NOT_0 "1"
#define LWS_NOT_1 "0"
template <int HALF, int SWZ, int NEGLO, int NEGHI> __device__ __forceinline__ v2f pk_fma_w(v2f acc, wp_t w, v2f v) {
    if constexpr (HALF == 0 && SWZ == 0 && NEGLO == 0 && NEGHI == 0) LWS_PKFMA(0, 0, 0, 0);
    else if constexpr (HALF == 0 && SWZ == 0 && NEGLO == 1 && NEGHI == 1) LWS_PKFMA(0, 0, 1, 1);
    else if constexpr (HALF == 1 && SWZ == 0 && NEGLO == 0 && NEGHI == 0) LWS_PKFMA(1, 0, 0, 0);
    else if constexpr (HALF == 1 && SWZ == 0 && NEGLO == 1 && NEGHI == 1) LWS_PKFMA(1, 0, 1, 1);
    else if constexpr (HALF == 0 && SWZ == 1 && NEGLO == 1 && NEGHI == 0) LWS_PKFMA(0, 1, 1, 0);
    else if constexpr (HALF == 0 && SWZ == 1 && NEGLO == 0 && NEGHI == 1) LWS_PKFMA(0, 1, 0, 1);
    else if constexpr (HALF == 1 && SWZ == 1 && NEGLO == 1 && NEGHI == 0) LWS_PKFMA(1, 1, 1, 0);
    else if constexpr (HALF == 1 && SWZ == 1 && NEGLO == 0 && NEGHI == 1) LWS_PKFMA(1, 1, 0, 1);
    else static_assert(HALF < 0, "modifier combination not instantiated");
    return acc;
}
// x - y in one instruction (the compiler splits a vector subtraction into two scalar ones)
__device__ __forceinline__ v2f pk_sub(v2f x, v2f y) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(y));
    return d;
}
// x * j^ROT + y
template <int ROT> __device__ __forceinline__ v2f pk_add_rot(v2f x, v2f y) {
    v2f d;
    if constexpr ((ROT & 3) == 0) d = x + y;
    else if constexpr ((ROT & 3) == 1) asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]" : "=v"(d) : "v"(x), "v"(y));
    else if constexpr ((ROT & 3) == 2) d = pk_sub(y, x);
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(d) : "v"(x), "v"(y));
    return d;
}

// acc += w*b + conj(w)*c with w = (wr, wi) * j^ROT   (grouped form of lwslib.cpp:310-311)
//   = p (b + c) + q j (b - c)   with (p, q) = (wr, wi), (-wi, wr), (-wr, -wi), (wi, -wr) for ROT = 0..3,
//   j (dx, dy) = (-dy, dx)
// One asm statement per group of instructions: the compiler cannot see inside a statement and puts a wait state
// (s_nop) after each one whose result the next instruction reads -- a quarter of the kernel's s_nops before the
// statements were merged.  (The packed fp32 operations have no such forwarding hazard; the rule is for 16-bit
// destination selects.)  Operand modifiers of the two multiply-adds for ROT = 0..3, see pk_fma_w:
#define LWS_M1_0 "op_sel:[0,0,0] op_sel_hi:[0,1,1]"
#define LWS_M2_0 "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
#define LWS_M1_1 "op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
#define LWS_M2_1 "op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]"
#define LWS_M1_2 "op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
#define LWS_M2_2 "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]"
#define LWS_M1_3 "op_sel:[1,0,0] op_sel_hi:[1,1,1]"
#define LWS_M2_3 "op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]"
#define LWS_NEG2 " neg_lo:[0,1] neg_hi:[0,1]"
// acc += p (b + c) + q j (b - c)
#define LWS_PAIR_ASM(M1, M2)                                                                             \
    asm("v_pk_add_f32 %[s], %[b], %[c]\n\t"                                                               \
        "v_pk_add_f32 %[d], %[b], %[c]" LWS_NEG2 "\n\t"                                                   \
        "v_pk_fma_f32 %[a], %[w], %[s], %[a] " M1 "\n\t"                                                  \
        "v_pk_fma_f32 %[a], %[w], %[d], %[a] " M2                                                          \
        : [a] "+v"(acc), [s] "=&v"(t0), [d] "=&v"(t1)                                                      \
        : [w] LWS_WREG(w), [b] "v"(vb), [c] "v"(vc))
// b = um +- dp, c = dm +- up, then the same
#define LWS_SWPJ " op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"   // x + j y
#define LWS_SWMJ " op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]"   // x - j y
#define LWS_QUAD_ASM(SGB, SGC, M1, M2)                                                                   \
    asm("v_pk_add_f32 %[b], %[um], %[dp]" SGB "\n\t"                                                      \
        "v_pk_add_f32 %[c], %[dm], %[up]" SGC "\n\t"                                                      \
        "v_pk_add_f32 %[s], %[b], %[c]\n\t"                                                               \
        "v_pk_add_f32 %[b], %[b], %[c]" LWS_NEG2 "\n\t"                                                   \
        "v_pk_fma_f32 %[a], %[w], %[s], %[a] " M1 "\n\t"                                                  \
        "v_pk_fma_f32 %[a], %[w], %[b], %[a] " M2                                                          \
        : [a] "+v"(acc), [b] "=&v"(t0), [c] "=&v"(t1), [s] "=&v"(t2)                                       \
        : [w] LWS_WREG(w), [um] "v"(vum), [up] "v"(vup), [dm] "v"(vdm), [dp] "v"(vdp))

// acc += w*b + conj(w)*c with w = (wr, wi) * j^ROT   (grouped form of lwslib.cpp:310-311)
//   = p (b + c) + q j (b - c)   with (p, q) = (wr, wi), (-wi, wr), (-wr, -wi), (wi, -wr) for ROT = 0..3,
//   j (dx, dy) = (-dy, dx)
template <int ROT> __device__ __forceinline__ void pair_rot(float2 &a, wp_t w, float2 b, float2 c) {
    v2f acc = vv(a), t0, t1;
    const v2f vb = vv(b), vc = vv(c);
    constexpr int R = ROT & 3;
    if constexpr (R == 0) LWS_PAIR_ASM(LWS_M1_0, LWS_M2_0);
    else if constexpr (R == 1) LWS_PAIR_ASM(LWS_M1_1, LWS_M2_1);
    else if constexpr (R == 2) LWS_PAIR_ASM(LWS_M1_2, LWS_M2_2);
    else LWS_PAIR_ASM(LWS_M1_3, LWS_M2_3);
    a = ff(acc);
}
// the four taps (m-R, c-k), (m-R, c+k), (m+R, c-k), (m+R, c+k) of one weight.  With tau = exp(2j pi mod R / Q) the twiddle of
// the bin, W[mod] = W0 tau multiplies (m-R, c-k) and W[-mod] = W0 conj(tau) multiplies (m+R, c+k) (conjugates for the other
// two).  tau = j^ROT (even eighth turns): b = um +- dp, c = dm +- up (minus for an odd quarter turn: W[-mod] = -W[mod]), then
// acc += w j^ROT b + conj(w j^ROT) c.  tau = j^ROT e^{j pi/4} (ODD, Q = 8): conj(tau) = tau rho with rho = -+j, so
// b = um + rho dp, c = dm + conj(rho) up, then the same with w = (W0 e^{j pi/4}) -- the second weight set.
template <int ROT, int ODD = 0> __device__ __forceinline__ void quad_rot(float2 &a, wp_t w, float2 um, float2 up, float2 dm, float2 dp) {
    v2f acc = vv(a), t0, t1, t2;
    const v2f vum = vv(um), vup = vv(up), vdm = vv(dm), vdp = vv(dp);
    constexpr int R = ROT & 3;
    if constexpr (ODD == 0) {
        if constexpr (R == 0) LWS_QUAD_ASM("", "", LWS_M1_0, LWS_M2_0);
        else if constexpr (R == 1) LWS_QUAD_ASM(LWS_NEG2, LWS_NEG2, LWS_M1_1, LWS_M2_1);
        else if constexpr (R == 2) LWS_QUAD_ASM("", "", LWS_M1_2, LWS_M2_2);
        else LWS_QUAD_ASM(LWS_NEG2, LWS_NEG2, LWS_M1_3, LWS_M2_3);
    } else {   // rho = (-1)^ROT (-j)
        if constexpr (R == 0) LWS_QUAD_ASM(LWS_SWMJ, LWS_SWPJ, LWS_M1_0, LWS_M2_0);
        else if constexpr (R == 1) LWS_QUAD_ASM(LWS_SWPJ, LWS_SWMJ, LWS_M1_1, LWS_M2_1);
        else if constexpr (R == 2) LWS_QUAD_ASM(LWS_SWMJ, LWS_SWPJ, LWS_M1_2, LWS_M2_2);
        else LWS_QUAD_ASM(LWS_SWPJ, LWS_SWMJ, LWS_M1_3, LWS_M2_3);
    }
    a = ff(acc);
}
// Frames m-+1 and m-+3 sharing one weight (FLAG_R13): b = um +- dp, c = dm +- up, then B = p3b j^RB + b,
// C = p3c j^-RB + c, then acc += w j^ROT B + conj(w j^ROT) C -- eight instructions in ONE statement: the compiler puts a
// wait state in front of every asm statement that reads the register the previous instruction wrote, and every
// instruction of a wave costs an issue slot whatever it does.
#define LWS_ROTX_0 ""
#define LWS_ROTX_1 " op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]"
#define LWS_ROTX_2 " neg_lo:[1,0] neg_hi:[1,0]"
#define LWS_ROTX_3 " op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]"
#define LWS_R13_ASM(SG, RB, RC, M1, M2)                                                                  \
    asm("v_pk_add_f32 %[b], %[um], %[dp]" SG "\n\t"                                                       \
        "v_pk_add_f32 %[c], %[dm], %[up]" SG "\n\t"                                                       \
        "v_pk_add_f32 %[b], %[pb], %[b]" RB "\n\t"                                                        \
        "v_pk_add_f32 %[c], %[pc], %[c]" RC "\n\t"                                                        \
        "v_pk_add_f32 %[s], %[b], %[c]\n\t"                                                               \
        "v_pk_add_f32 %[b], %[b], %[c]" LWS_NEG2 "\n\t"                                                   \
        "v_pk_fma_f32 %[a], %[w], %[s], %[a] " M1 "\n\t"                                                  \
        "v_pk_fma_f32 %[a], %[w], %[b], %[a] " M2                                                          \
        : [a] "+v"(acc), [b] "=&v"(t0), [c] "=&v"(t1), [s] "=&v"(t2)                                       \
        : [w] LWS_WREG(w), [um] "v"(vum), [up] "v"(vup), [dm] "v"(vdm), [dp] "v"(vdp), [pb] "v"(vpb), [pc] "v"(vpc))
#define LWS_R13_ROT(SG, M1, M2)                                                                          \
    do {                                                                                                 \
        if constexpr (RB == 0) LWS_R13_ASM(SG, LWS_ROTX_0, LWS_ROTX_0, M1, M2);                          \
        else if constexpr (RB == 1) LWS_R13_ASM(SG, LWS_ROTX_1, LWS_ROTX_3, M1, M2);                     \
        else if constexpr (RB == 2) LWS_R13_ASM(SG, LWS_ROTX_2, LWS_ROTX_2, M1, M2);                     \
        else LWS_R13_ASM(SG, LWS_ROTX_3, LWS_ROTX_1, M1, M2);                                            \
    } while (0)
template <int ROT, int RB_>
__device__ __forceinline__ void r13_rot(float2 &a, wp_t w, float2 um, float2 up, float2 dm, float2 dp, float2 pb, float2 pc) {
    v2f acc = vv(a), t0, t1, t2;
    const v2f vum = vv(um), vup = vv(up), vdm = vv(dm), vdp = vv(dp), vpb = vv(pb), vpc = vv(pc);
    constexpr int R = ROT & 3, RB = RB_ & 3;
    if constexpr (R == 0) LWS_R13_ROT("", LWS_M1_0, LWS_M2_0);
    else if constexpr (R == 1) LWS_R13_ROT(LWS_NEG2, LWS_M1_1, LWS_M2_1);
    else if constexpr (R == 2) LWS_R13_ROT("", LWS_M1_2, LWS_M2_2);
    else LWS_R13_ROT(LWS_NEG2, LWS_M1_3, LWS_M2_3);
    a = ff(acc);
}
// b = um +- dp, c = dm +- up (the partial sums rows 3 leave for rows 1) in one statement
template <int ODD> __device__ __forceinline__ void bc_pair(float2 &b, float2 &c, float2 um, float2 up, float2 dm, float2 dp) {
    v2f vb, vc;
    const v2f vum = vv(um), vup = vv(up), vdm = vv(dm), vdp = vv(dp);
    if constexpr (ODD)
        asm("v_pk_add_f32 %0, %2, %3" LWS_NEG2 "\n\tv_pk_add_f32 %1, %4, %5" LWS_NEG2 : "=&v"(vb), "=&v"(vc) : "v"(vum), "v"(vdp), "v"(vdm), "v"(vup));
    else
        asm("v_pk_add_f32 %0, %2, %3\n\tv_pk_add_f32 %1, %4, %5" : "=&v"(vb), "=&v"(vc) : "v"(vum), "v"(vdp), "v"(vdm), "v"(vup));
    b = ff(vb);
    c = ff(vc);
}
// the same for a weight whose imaginary part is exactly zero (W[0][r][0] of symmetric windows): half the work
#define LWS_REAL_ASM(SG, M)                                                                              \
    asm("v_pk_add_f32 %[t], %[b], %[c]" SG "\n\t"                                                         \
        "v_pk_fma_f32 %[a], %[w], %[t], %[a] " M                                                           \
        : [a] "+v"(acc), [t] "=&v"(t0)                                                                     \
        : [w] LWS_WREG(w), [b] "v"(vb), [c] "v"(vc))
template <int ROT> __device__ __forceinline__ void pair_rot_real(float2 &a, wp_t w, float2 b, float2 c) {
    constexpr int R = ROT & 3;
    v2f acc = vv(a), t0;
    const v2f vb = vv(b), vc = vv(c);
    // (wr, 0) j^ROT: p (b + c) for even ROT, q j (b - c) for odd ROT -- the first / second multiply-add of pair_rot
    if constexpr (R == 0) LWS_REAL_ASM("", LWS_M1_0);
    else if constexpr (R == 1) LWS_REAL_ASM(LWS_NEG2, LWS_M2_1);
    else if constexpr (R == 2) LWS_REAL_ASM("", LWS_M1_2);
    else LWS_REAL_ASM(LWS_NEG2, LWS_M2_3);
    a = ff(acc);
}
// ---- LWS_TW (per-lane twiddles) ----------------------------------------------------------------------------------
// acc += w b + conj(w) c with the weight in a VGPR pair: the twiddle of the lane's bin
__device__ __forceinline__ void pair_v(float2 &a, float2 wv, float2 b, float2 c) {
    v2f acc = vv(a), t0, t1;
    const v2f vb = vv(b), vc = vv(c), w = vv(wv);
    asm("v_pk_add_f32 %[s], %[b], %[c]\n\t"
        "v_pk_add_f32 %[d], %[b], %[c]" LWS_NEG2 "\n\t"
        "v_pk_fma_f32 %[a], %[w], %[s], %[a] " LWS_M1_0 "\n\t"
        "v_pk_fma_f32 %[a], %[w], %[d], %[a] " LWS_M2_0
        : [a] "+v"(acc), [s] "=&v"(t0), [d] "=&v"(t1)
        : [w] "v"(w), [b] "v"(vb), [c] "v"(vc));
    a = ff(acc);
}
// acc += w v (CONJ = 0) or conj(w) v (CONJ = 1), w a base weight
template <int CONJ> __device__ __forceinline__ void cmul_w(float2 &a, wp_t w, float2 v) {
    v2f acc = vv(a);
    const v2f x = vv(v);
    if constexpr (CONJ == 0)
        asm("v_pk_fma_f32 %[a], %[w], %[v], %[a] " LWS_M1_0 "\n\t"
            "v_pk_fma_f32 %[a], %[w], %[v], %[a] op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
            : [a] "+v"(acc) : [w] LWS_WREG(w), [v] "v"(x));
    else
        asm("v_pk_fma_f32 %[a], %[w], %[v], %[a] " LWS_M1_0 "\n\t"
            "v_pk_fma_f32 %[a], %[w], %[v], %[a] op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]"
            : [a] "+v"(acc) : [w] LWS_WREG(w), [v] "v"(x));
    a = ff(acc);
}
// w v (CONJ = 0) or conj(w) v (CONJ = 1) without an accumulator to zero first: one multiply, one multiply-add
template <int CONJ> __device__ __forceinline__ float2 cmul_w_init(wp_t w, float2 v) {
    v2f r;
    const v2f x = vv(v);
    if constexpr (CONJ == 0)
        asm("v_pk_mul_f32 %[r], %[w], %[v] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
            "v_pk_fma_f32 %[r], %[w], %[v], %[r] op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
            : [r] "=&v"(r) : [w] LWS_WREG(w), [v] "v"(x));
    else
        asm("v_pk_mul_f32 %[r], %[w], %[v] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
            "v_pk_fma_f32 %[r], %[w], %[v], %[r] op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]"
            : [r] "=&v"(r) : [w] LWS_WREG(w), [v] "v"(x));
    return ff(r);
}
__device__ __forceinline__ float2 cmulf(float2 p, float2 q) { return make_float2(p.x * q.x - p.y * q.y, p.x * q.y + p.y * q.x); }
__device__ __forceinline__ float2 wp_value(wp_t w) { return make_float2(__uint_as_float((unsigned)(w & 0xffffffffull)), __uint_as_float((unsigned)(w >> 32))); }
__device__ __forceinline__ float2 cadd(float2 p, float2 q) { return ff(vv(p) + vv(q)); }
__device__ __forceinline__ float2 csub(float2 p, float2 q) { return ff(pk_sub(vv(p), vv(q))); }

// eighth turns of the twiddle exp(2j pi mod R / Q) of frame pair R at a bin with bin % Q = mod (even unless Q = 8)
template <int Q> __host__ __device__ constexpr int eighths(int mod, int R) { return ((mod * R) % Q) * (8 / Q); }
// where W[0][R][k] (set 0) / W[0][R][k] exp(j pi / 4) (set 1: the odd eighth turns of Q = 8) sits in the weight registers of
// the wave that sums frames m-+R.  Q = 8: position in that wave's list -- its frames in ascending order, set 0 then (odd
// frames only) set 1
template <int Q, int L> __host__ __device__ constexpr int widx(int set, int R, int k) {
    if (!LWS_Q8) return (set * Q + R) * (L + 1) + k;
    int idx = 0;
    for (int r = 0; r < Q; ++r) {
        if (row_owner(r) != row_owner(R)) continue;
        for (int st = 0; st < 2; ++st) {
            if (st == 1 && (r & 1) == 0) continue;
            if (r == R && st == set) return idx + k;
            idx += L + 1;
        }
    }
    return -1;
}

// Contribution of the centre frame (W[.,0,k] does not depend on bin % Q) to the bin at phase PH / clock PB.
template <int L, uint64_t MASK, int PH, int PB, int RE = 0>
__device__ __forceinline__ void centre_sum(const SysArgs &a, const LaneCtx &cx, bool st, bool en, float2 self_old,
                                           float2 next_old, float2 prev_out, float2 &acc) {
    static_for<L>([&](auto ik) {
        constexpr int k = decltype(ik)::value + 1;
        if constexpr ((MASK >> k) & 1ull) {
            float2 lo = tap_any<PH, PB, 0, -k, 0>(cx, self_old, next_old, prev_out);
            float2 hi = tap_any<PH, PB, 0, k, 0>(cx, self_old, next_old, prev_out);
            // images, branch-free: the alternative source is fetched by every lane (the address is valid for all of
            // them) and selected for the one lane at the frame edge
            if constexpr (PH - k < 0) {        // first bins of a frame: (m, c-k) is the image of bin k-c
                const float2 im = tap_any<PH, PB, 0, -k, 1>(cx, self_old, next_old, prev_out);
                lo.x = st ? im.x : lo.x; lo.y = st ? im.y : lo.y;
            }
            if constexpr (PH < th0(RE) && PH + k >= th0(RE)) {   // last bins of a frame: Nyquist bin or an image
                const float2 im = tap_any<PH, PB, 0, k, 1 + th0(RE) - PH>(cx, self_old, next_old, prev_out);
                hi.x = en ? im.x : hi.x; hi.y = en ? im.y : hi.y;
            }
            if constexpr (RE != 0 && PH + k >= th1(RE)) {        // the same from the block before the end block
                const float2 im = tap_any<PH, PB, 0, k, 1 + th1(RE) - PH>(cx, self_old, next_old, prev_out);
                hi.x = cx.is_end1 ? im.x : hi.x; hi.y = cx.is_end1 ? im.y : hi.y;
            }
            pair_rot<0>(acc, a.w[widx<8, L>(0, 0, k)], lo, hi);   // (the centre frame's weights come first in every build)
        }
    });
}

// Structure flags carried in the top bits of the MASK template word (the tap mask itself needs Q*(L+1) <= 48 bits)
constexpr uint64_t FLAG_K0REAL = 1ull << 62;  // Im W[0][r][0] == 0 for r >= 1
constexpr uint64_t FLAG_R13 = 1ull << 63;     // Q = 4 and W[0][3][k] == j^k W[0][1][k] for 2 <= k <= L (sqrt-Hann, 75% overlap)

// With FLAG_R13 the taps k >= 2 of frames m-+3 share the weight of frames m-+1 up to a quarter turn:
//   W1 (j^mod B1 + j^(k-mod) B3) + conj(W1) (j^-mod C1 + j^(mod-k) C3),   B = um +- dp, C = dm +- up,
// so rows 3 only contribute rotated partial sums (kept in P3) and rows 1 do the complex multiply for both.
template <int L> struct R13Partials { float2 b[L + 1], c[L + 1]; };

// one group of taps (the four taps |dk| = K of frames m-R and m+R, or the two taps dk = 0) of the bin at phase PH;
// the bin sits at index L + 1 + OFFS of the tap windows tu (frame m-R) / td (frame m+R), see load_cells
template <int Q, int L, uint64_t MASK, int PH, int R, int OFFS, int K, int N>
__device__ __forceinline__ void rows_group(const SysArgs &a, const float2 (&tu)[N], const float2 (&td)[N],
                                           R13Partials<L> &p3, float2 &accr) {
    constexpr int K1 = L + 1;
    constexpr int mod = PH % Q;
    constexpr int e8 = eighths<Q>(mod, R), rot = e8 >> 1, odd = e8 & 1;  // exp(2j*pi*mod*R/Q) = j^rot (e^{j pi/4})^odd
    constexpr bool r13 = (MASK & FLAG_R13) != 0 && Q == 4;
    static_assert(L + 1 + OFFS + K < N, "tap window too short");
    // accr: the bin's running sum (every group of taps is added to it directly: no partial sums to zero and combine)
    if constexpr (((MASK >> (R * K1 + K)) & 1ull) == 0) {
        return;
    } else if constexpr (K == 0) {
        if constexpr ((MASK & FLAG_K0REAL) != 0 && !odd) pair_rot_real<rot>(accr, a.w[widx<Q, L>(0, R, 0)], tu[L + 1 + OFFS], td[L + 1 + OFFS]);
        else pair_rot<rot>(accr, a.w[widx<Q, L>(odd, R, 0)], tu[L + 1 + OFFS], td[L + 1 + OFFS]);
    } else {
        constexpr int k = K;
        // W[mod]*S[m-r,c-k] + conj(W[mod])*S[m+r,c-k] + W[-mod]*S[m+r,c+k] + conj(W[-mod])*S[m-r,c+k]
        // with W[-mod] = +-W[mod] for a real / imaginary twiddle (the LWSQ2 / LWSQ4 grouping)
        const wp_t w = a.w[widx<Q, L>(odd, R, k)];
        const float2 um = tu[L + 1 - k + OFFS], up = tu[L + 1 + k + OFFS], dm = td[L + 1 - k + OFFS], dp = td[L + 1 + k + OFFS];
        if constexpr (!(r13 && (R == 1 || R == 3) && k >= 2)) {
            quad_rot<rot, odd>(accr, w, um, up, dm, dp);
        } else if constexpr (R == 3) {
            bc_pair<(rot & 1)>(p3.b[k], p3.c[k], um, up, dm, dp);      // rotated when rows 1 pick them up
        } else {
            // j^rot1 (B1 + j^(k+rot3-rot1) B3) and j^-rot1 (C1 + j^(rot1-k-rot3) C3): one multiply for both rows
            constexpr int rot3 = eighths<Q>(mod, 3) >> 1;
            r13_rot<rot, (k + rot3 - rot + 8) & 3>(accr, w, um, up, dm, dp, p3.b[k], p3.c[k]);
        }
    }
}
// Contribution of frames m-R and m+R to the bin at phase PH; OFFS = 0 .. 3: which bin of the quad
template <int Q, int L, uint64_t MASK, int PH, int R, int OFFS, int N>
__device__ __forceinline__ void rows_sum(const SysArgs &a, const float2 (&tu)[N], const float2 (&td)[N],
                                         R13Partials<L> &p3, float2 &accr) {
    static_for<L + 1>([&](auto ik) { rows_group<Q, L, MASK, PH, R, OFFS, decltype(ik)::value>(a, tu, td, p3, accr); });
}

// ---- two pairs of bins from one set of tap windows: the quads (0,1) + (2,3) and (4,5) + (6,7) of a block ------------------
// Pairs start on even bins, so a pair never straddles two frames of a lane (a block of 8 bins lies in one frame) and a ring
// cell -- (even time, odd time) -- holds the two outputs of one pair.  The taps the four bins PA0 .. PA0+3 need from a
// neighbour frame are the bins PA0-L .. PA0+3+L: L+3 cells, of the first and the last of which one half is used (8-byte
// reads).  The first pair of a quad therefore also sums the neighbour-frame taps of the second pair's bins (the sums do not
// depend on anything the first pair produces) and hands them over in registers.
// Two frames cannot deliver their last (half) cell yet when the first pair starts -- frame m-1 (its bins are 8 steps ahead
// of this lane's: the cell holds what it produces during this very pair) and frame m+LATE_DN (previous sweep, LAG - 8 LATE_DN
// = 8 steps ahead).  One tap group of each touches it: (fourth bin, k = L).  Its other operands are handed over as well and
// the second pair finishes it after fetching the half cell.  (L = 3: nothing is late.)
template <int L> struct QuadCarry {
    float2 accA, accB;                               // neighbour-frame sums of the second pair's bins so far
    float2 um1, dm1, dp1, um3, c3;                   // the unfinished group with FLAG_R13: the operands of r13_rot() that are known
    float2 g[QMAX][4];                               // kernels without FLAG_R13: [R] the operands (um, up, dm, dp) of quad_rot()
};
// time of the last half cell of the window of frame m+DR, relative to the start of the quad's first pair: not produced yet?
template <int DR, int L> __host__ __device__ constexpr bool quad_late_frame() { return SKEW * DR - (DR > 0 ? LAG : 0) + L + 3 >= 0; }
template <int OFFS, int K, int L> __host__ __device__ constexpr bool quad_deferred() { return OFFS == 3 && K == L; }

// neighbour-frame taps of a bin of the SECOND pair (OFFS = 2, 3), summed during the first pair
template <int Q, int L, uint64_t MASK, int PH, int R, int OFFS, int N>
__device__ __forceinline__ void rows_sum_ahead(const SysArgs &a, const float2 (&tu)[N], const float2 (&td)[N],
                                               R13Partials<L> &p3, float2 &accr, QuadCarry<L> &qc) {
    constexpr int K1 = L + 1;
    constexpr int rot = eighths<Q>(PH % Q, R) >> 1;
    static_for<L + 1>([&](auto ik) {
        constexpr int k = decltype(ik)::value;
        constexpr bool r13 = (MASK & FLAG_R13) != 0 && Q == 4;
        constexpr bool late = quad_deferred<OFFS, k, L>() && (quad_late_frame<-R, L>() || quad_late_frame<R, L>());
        constexpr int c = L + 1 + OFFS;               // the bin's place in the windows
        if constexpr (!late) {
            rows_group<Q, L, MASK, PH, R, OFFS, k>(a, tu, td, p3, accr);
        } else if constexpr (((MASK >> (R * K1 + k)) & 1ull) == 0) {
        } else if constexpr (!r13) {          // the late operand is filled in by the second pair (quad_finish_plain)
            qc.g[R][0] = tu[c - k];
            if constexpr (!quad_late_frame<-R, L>()) qc.g[R][1] = tu[c + k];
            qc.g[R][2] = td[c - k];
            if constexpr (!quad_late_frame<R, L>()) qc.g[R][3] = td[c + k];
        } else {
            static_assert(k >= 2 && (R == 1 || R == 3), "only the shared-weight groups reach the late cells");
            if constexpr (R == 3) {          // frame m+3's tap at +k is late: keep um3 and c3 = dm3 +- up3
                qc.um3 = tu[c - k];
                if constexpr ((rot & 1) == 0) qc.c3 = cadd(td[c - k], tu[c + k]);
                else qc.c3 = csub(td[c - k], tu[c + k]);
            } else {                          // frame m-1's tap at +k is late: keep the other three
                qc.um1 = tu[c - k];
                qc.dm1 = td[c - k];
                qc.dp1 = td[c + k];
            }
        }
    });
}
// the unfinished group of a kernel without FLAG_R13: `late` is the tap that was not there yet (frame m-R's or m+R's, at +L)
template <int Q, int L, uint64_t MASK, int PH, int R>
__device__ __forceinline__ void quad_finish_plain(const SysArgs &a, const QuadCarry<L> &qc, float2 late, float2 &accr, float2 late_dn = make_float2(0.f, 0.f)) {
    // (late_dn: with a 16-step lag both frames of the pair R = 1 are late -- m-1 delivers `late`, m+1 `late_dn`)
    constexpr int K1 = L + 1;
    constexpr int e8 = eighths<Q>(PH % Q, R), rot = e8 >> 1, odd = e8 & 1;
    if constexpr ((MASK >> (R * K1 + L)) & 1ull)
        quad_rot<rot, odd>(accr, a.w[widx<Q, L>(odd, R, L)], qc.g[R][0], quad_late_frame<-R, L>() ? late : qc.g[R][1], qc.g[R][2],
                           quad_late_frame<R, L>() ? ((quad_late_frame<-R, L>() && LATE_DN == R) ? late_dn : late) : qc.g[R][3]);
}
// the unfinished group with FLAG_R13, by the second pair: up1 = frame m-1's late tap, dp3 = frame m+3's
template <int Q, int L, uint64_t MASK, int PH>
__device__ __forceinline__ void quad_finish(const SysArgs &a, const QuadCarry<L> &qc, float2 up1, float2 dp3, float2 &accr) {
    constexpr int K1 = L + 1, mod = PH % Q;
    constexpr int rot1 = eighths<Q>(mod, 1) >> 1, rot3 = eighths<Q>(mod, 3) >> 1;
    if constexpr ((MASK >> (1 * K1 + L)) & 1ull) {
        float2 b3;
        if constexpr ((rot3 & 1) == 0) b3 = cadd(qc.um3, dp3);
        else b3 = csub(qc.um3, dp3);
        r13_rot<rot1, (L + rot3 - rot1 + 8) & 3>(accr, a.w[1 * K1 + L], qc.um1, up1, qc.dm1, qc.dp1, b3, qc.c3);
    }
}

// the twiddles of the four bins of a quad for frame pair R, fetched together with the tap windows (not when the sums are done)
// (64-step-ring build, whose waves have registers to spare and a SIMD nearly to themselves: -5 %; the other builds read each twiddle where
// it is used: +1 % otherwise)
template <int PH0, int R> __device__ __forceinline__ void tw_fetch(int twbase, float2 (&tau)[4]) {
    if constexpr (LWS_Q8) {
#pragma unroll
        for (int o = 0; o < 4; ++o) tau[o] = lds_read(twbase + (PH0 + o) * TW_ROW + (R - 1) * 8);
    } else {
#pragma unroll
        for (int o = 0; o < 4; ++o) tau[o] = make_float2(__int_as_float(twbase + (PH0 + o) * TW_ROW + (R - 1) * 8), 0.f);   // the address, for tw_value
    }
}
__device__ __forceinline__ float2 tw_value(float2 tau) { return LWS_Q8 ? tau : lds_read(__float_as_int(tau.x)); }
// LWS_TW: contribution of frames m-R and m+R to the bin at phase PH (bin OFFS of the quad), twiddle from the table:
//   tau U + conj(tau) D,   U = sum_k W0[R][k] S[m-R,c-k] + conj(W0[R][k]) S[m-R,c+k],   D = sum_k W0[R][k] S[m+R,c+k] + conj(W0[R][k]) S[m+R,c-k]
// (lwslib.cpp:321-352 with W[mod] = W0 tau, W[modneg] = W0 conj(tau)).  The one group whose last operand is not in the ring yet
// (fourth bin, k = L, frames m-1 / m+LATE_DN) is left to the second pair as in the other builds (QuadCarry::g, tw_finish).
template <int Q, int L, uint64_t MASK, int PH, int R, int OFFS, int N>
__device__ __forceinline__ void tw_rows(const SysArgs &a, float2 tau, const float2 (&tu)[N], const float2 (&td)[N],
                                        float2 &accr, QuadCarry<L> &qc) {
    constexpr int K1 = L + 1, c = L + 1 + OFFS;
    static_assert(c + L < N && PH >= 0 && PH < 8, "tap window too short");
    // (the tap k = 0 starts the two sums: nothing to zero)
    constexpr bool k0 = ((MASK >> (R * K1)) & 1ull) != 0;
    float2 U, D;
    if constexpr (k0) {
        const wp_t w0 = a.w[widx<Q, L>(0, R, 0)];
        U = cmul_w_init<0>(w0, tu[c]);
        D = cmul_w_init<1>(w0, td[c]);
    } else {
        U = make_float2(0.f, 0.f);
        D = make_float2(0.f, 0.f);
    }
    static_for<L>([&](auto ik) {
        constexpr int k = decltype(ik)::value + 1;
        if constexpr (((MASK >> (R * K1 + k)) & 1ull) != 0) {
            const wp_t w = a.w[widx<Q, L>(0, R, k)];
            if constexpr (quad_deferred<OFFS, k, L>() && (quad_late_frame<-R, L>() || quad_late_frame<R, L>())) {
                qc.g[R][0] = tu[c - k];
                if constexpr (!quad_late_frame<-R, L>()) qc.g[R][1] = tu[c + k];
                qc.g[R][2] = td[c - k];
                if constexpr (!quad_late_frame<R, L>()) qc.g[R][3] = td[c + k];
            } else {
                pair_rot<0>(U, w, tu[c - k], tu[c + k]);
                pair_rot<0>(D, w, td[c + k], td[c - k]);
            }
        }
    });
    pair_v(accr, tw_value(tau), U, D);
}
// ... and the deferred group, by the second pair: `late` is the tap that was not there yet (frame m-R's or m+R's, at +L)
template <int Q, int L, uint64_t MASK, int PH, int R>
__device__ __forceinline__ void tw_finish(const SysArgs &a, const LaneCtx &cx, const QuadCarry<L> &qc, float2 late, float2 &accr) {
    constexpr int K1 = L + 1;
    if constexpr ((MASK >> (R * K1 + L)) & 1ull) {
        const wp_t w = a.w[widx<Q, L>(0, R, L)];
        float2 U = make_float2(0.f, 0.f), D = make_float2(0.f, 0.f);
        pair_rot<0>(U, w, qc.g[R][0], quad_late_frame<-R, L>() ? late : qc.g[R][1]);
        pair_rot<0>(D, w, quad_late_frame<R, L>() ? late : qc.g[R][3], qc.g[R][2]);
        pair_v(accr, lds_read(cx.tw + PH * TW_ROW + (R - 1) * 8), U, D);
    }
}

// Cells C0 .. C0+NC-1 of the taps frame m+DR contributes to the quad that starts at phase PA0 (0 or 4): t[j] is the tap at
// block-relative bin PA0 - L - 1 + j, so bin PA0 + OFFS sits at t[L + 1 + OFFS] and cell i = (t[2i], t[2i+1]) is one aligned
// 16-byte ring cell (even time, odd time).  The window is t[1] .. t[2L+4]: of cells 0 and L+2 only one half is read.
// Cells are (even bin, odd bin) pairs, so a cell lies entirely inside the frame or entirely among its images; for the one
// lane of a wave at the start / end of its frame the image cells come from the pseudo-lanes (LaneCtx::wlo / whi).
// CO: clock of the block the quad belongs to, relative to the block `cx` addresses (0, or 8 for a helper wave working on the
// next block)
// KMASK: bit k set <=> the taps +-k of this frame pair are used (a tap nobody uses is not fetched: the ring reads are volatile,
// and a fetch whose result is dropped still makes the next user of its register wait for it)
template <int L> __host__ __device__ constexpr bool window_slot_used(uint32_t kmask, int j) {
    for (int offs = 0; offs < 4; ++offs) {
        const int d = j - (L + 1 + offs), k = d < 0 ? -d : d;
        if (k <= L && ((kmask >> k) & 1u)) return true;
    }
    return false;
}
template <int PA0, int DR, int L, int C0, int NC, uint32_t KMASK, int CO = 0, int RE = 0, int N = 0>
__device__ __forceinline__ void load_cells(const LaneCtx &cx, float2 (&t)[N]) {
    static_assert((PA0 & 3) == 0 && (L & 1) == 1 && C0 + NC <= L + 3 && N >= 2 * L + 5, "cell window");
    constexpr int base_off = SKEW * DR - (DR > 0 ? LAG : 0);
    constexpr int q_first = PA0 + CO + base_off - L - 1;         // even
    static_for<NC>([&](auto ip) {
        constexpr int i = C0 + decltype(ip)::value, j = 2 * i;
        constexpr int q = q_first + j;
        static_assert(q >= -RING && q + 1 <= 15, "ring retention exceeded");
        constexpr int b = PA0 - L - 1 + j;                       // the cell's even bin
        // (RE != 0, see th0: image cells start at bin C wherever it sits in the lane's block or the next one)
        constexpr bool img_lo = (b + 1 < 0), img_hi = (b >= th0(RE)) && (RE == 0 || b < th1(RE)), img_hi1 = (RE != 0) && (b >= th1(RE));
        constexpr int fl = floor_div8(q);
        constexpr int m = blk_mod(-fl);
        constexpr int within = q - 8 * fl;                       // even
        constexpr int setoff = (DR < 0) ? SET_BYTES : 0;
        int base = cx.ob[m];
        if constexpr (img_lo) base = cx.ob[m] + cx.wlo[DR + HALO];
        if constexpr (img_hi) base = cx.ob[m] + cx.whi[DR + HALO];
        if constexpr (img_hi1) base = cx.ob[m] + cx.whi1[DR + HALO];
        const int addr = base + (HALO + DR) * LANE_B + setoff + (within >> 1) * PAIR_BYTES;
        constexpr bool need0 = i != 0 && window_slot_used<L>(KMASK, j), need1 = i != L + 2 && window_slot_used<L>(KMASK, j + 1);
        if constexpr (need0 && need1) {
            const v4f v = lds_read128(addr);
            t[j] = make_float2(v.x, v.y);
            t[j + 1] = make_float2(v.z, v.w);
        } else if constexpr (need0) {
            t[j] = lds_read(addr);
        } else if constexpr (need1) {
            t[j + 1] = lds_read(addr + 8);
        }
    });
}

// Magnitude re-projection (lwslib.cpp:356-360): keep the old value unless the bin is active and the sum is
// non-zero.  target/|acc| is evaluated as target * rsqrt(|acc|^2) with the hardware reciprocal square root (1 ulp:
// the new magnitude is within 3e-7 relative of the target, tests/test_gpu_parity.py checks 1e-6).
__device__ __forceinline__ float2 project(float2 acc, float target, bool active, float2 old) {
    // (the weights carry a per-spectrogram power-of-two scale that keeps |acc|^2 inside the fp32 range -- k_systolic;
    // a sum more than 1e-19 below the spectrogram's largest magnitude counts as zero)
    const float m2 = acc.x * acc.x + acc.y * acc.y;
    const bool ok = active && (m2 > 0.f);
    float r = __frsqrt_rn(m2);
    const float sc = target * r;
    return ok ? make_float2(acc.x * sc, acc.y * sc) : old;
}

// Registers a compute lane carries from pair to pair: the previous sweep's values of its next four bins and its last output.
struct Carry { float2 o0, o1, o2, o3, prev_out; };

#define LWS_SETPRIO(n) asm volatile("s_setprio " #n)
__device__ __forceinline__ void lds_write128(int addr, float2 p, float2 q) {
    using lds_v4w = volatile __attribute__((address_space(3))) v4f;
    *(lds_v4w *)(unsigned)addr = (v4f){p.x, p.y, q.x, q.y};
}

// One pair of bins (phases PA even, PA+1) of one lane.
template <int Q, int L, uint64_t MASK, int PA, bool H16, int RE = 0>
__device__ __forceinline__ void compute_pair(const SysArgs &a, const LaneCtx &cx, Carry &cr, const float (&amp_cur)[8],
                                             QuadCarry<L> &qc) {
    static_assert((PA & 1) == 0, "pairs start on even bins");
    constexpr int K1 = L + 1;
    constexpr int PHB = PA + 1;                              // second bin
    constexpr int PA0 = PA & ~2;                             // first phase of the quad
    constexpr bool quad_first = (PA & 2) == 0;
    const bool st = cx.is_start, en = cx.is_end;
    // previous-sweep values of the own bins of the pair after the next one (ages LAG-4 and LAG-5 now): one ring cell
    const v4f o45 = lds_read128(ring_addr<PA, 4 - LAG>(cx.ob));
    // Issue priority (s_setprio; the SIMD serves the higher priority first, the older wave among equals).  The two sweep
    // slots of a SIMD run the same pair at the same time and every pair ends in a rendez-vous, so what counts is when
    // the LATER of the two publishes.  Left alone, the older wave wins every issue slot, finishes early and waits while
    // the younger one runs the second half of its pair alone at single-wave pace.  Raising the priority for the tail of
    // the pair (the serial part: last taps, two projections, the publish) and lowering it for the bulk in between lets
    // whichever wave is in its tail go first and the two leapfrog through the bulk: 41.8 -> 39.7 ms.  The service wave,
    // which every slot meets at every pair, stays above all of them.
    LWS_SETPRIO(1);
    float2 accA = make_float2(0.f, 0.f);
    float2 accB = make_float2(0.f, 0.f);
    if constexpr (NHELP > 0) {   // what the helper waves of this slot summed for this pair (frames row_owner() gives them)
        static_for<NHELP>([&](auto ih) {
            const v4f mb = lds_read128(cx.mbox + mbox_addr(0, decltype(ih)::value + 1, PA >> 1));
            accA = cadd(accA, make_float2(mb.x, mb.y));
            accB = cadd(accB, make_float2(mb.z, mb.w));
        });
    }
    centre_sum<L, MASK, PA, PA, RE>(a, cx, st, en, cr.o0, cr.o1, cr.prev_out, accA);
    // frame pairs m-+R.  With FLAG_R13 rows 3 leave partial sums for rows 1: order 2, 3, 1 keeps them short-lived.
    constexpr bool r13 = (MASK & FLAG_R13) != 0 && Q == 4;
    if constexpr (TW && quad_first) {
        // (LWS_TW: the same windows; per frame pair the sums U, D of each of the quad's four bins, turned by the bin's twiddle)
        static_assert(LATE_DN != 1 && !r13, "32- or 64-step ring");
        qc.accA = make_float2(0.f, 0.f);
        qc.accB = make_float2(0.f, 0.f);
        static_for<Q - 1>([&](auto ir) {
            constexpr int R = decltype(ir)::value + 1;
            constexpr uint32_t kmask = (uint32_t)((MASK >> (R * K1)) & ((1ull << K1) - 1ull));
            if constexpr (row_owner(R) == 0 && kmask != 0) {   // (else: a helper wave's frames, or frames this plan's Q does not have)
                float2 tu[2 * L + 6], td[2 * L + 6];
                load_cells<PA0, -R, L, 0, (quad_late_frame<-R, L>() ? L + 2 : L + 3), kmask, 0, RE>(cx, tu);
                load_cells<PA0, R, L, 0, (quad_late_frame<R, L>() ? L + 2 : L + 3), kmask, 0, RE>(cx, td);
                float2 tau[4];
                tw_fetch<PA, R>(cx.tw, tau);
                tw_rows<Q, L, MASK, PA, R, 0>(a, tau[0], tu, td, accA, qc);
                tw_rows<Q, L, MASK, PA + 1, R, 1>(a, tau[1], tu, td, accB, qc);
                tw_rows<Q, L, MASK, PA + 2, R, 2>(a, tau[2], tu, td, qc.accA, qc);
                tw_rows<Q, L, MASK, PA + 3, R, 3>(a, tau[3], tu, td, qc.accB, qc);
            }
            if constexpr (R == 1) LWS_SETPRIO(0);
            if constexpr (R == Q - 1) LWS_SETPRIO(2);
        });
    } else if constexpr (TW) {
        accA = cadd(accA, qc.accA);
        accB = cadd(accB, qc.accB);
        if constexpr (quad_late_frame<-1, L>()) {
            float2 u1[2 * L + 6], d3[2 * L + 6];
            load_cells<PA0, -1, L, L + 2, 1, (uint32_t)((MASK >> K1) & ((1ull << K1) - 1ull)), 0, RE>(cx, u1);
            tw_finish<Q, L, MASK, PHB, 1>(a, cx, qc, u1[2 * L + 4], accB);
            if constexpr (Q > LATE_DN && ((MASK >> ((Q > LATE_DN ? LATE_DN : 0) * K1)) & ((1ull << K1) - 1ull)) != 0) {
                load_cells<PA0, LATE_DN, L, L + 2, 1, (uint32_t)((MASK >> ((Q > LATE_DN ? LATE_DN : 0) * K1)) & ((1ull << K1) - 1ull)), 0, RE>(cx, d3);
                tw_finish<Q, L, MASK, PHB, (Q > LATE_DN ? LATE_DN : 1)>(a, cx, qc, d3[2 * L + 4], accB);
            }
        }
        LWS_SETPRIO(2);
    } else if constexpr (quad_first) {
        // this pair and the neighbour-frame sums of the next one, from one set of windows (rows_sum_ahead)
        R13Partials<L> p3A, p3B, p3C, p3D;
        qc.accA = make_float2(0.f, 0.f);
        qc.accB = make_float2(0.f, 0.f);
        static_for<Q - 1>([&](auto ir) {
            constexpr int i = decltype(ir)::value;
            constexpr int R = r13 ? (i == 0 ? 2 : (i == 1 ? 3 : 1)) : i + 1;
            if constexpr (row_owner(R) == 0) {   // (else: summed by a helper wave, in accA / accB already)
                float2 tu[2 * L + 6], td[2 * L + 6];
                static_assert(quad_late_frame<-1, L>() == quad_late_frame<LATE_DN, L>() && !quad_late_frame<-2, L>() &&
                              (LATE_DN == 1 || (!quad_late_frame<LATE_DN - 1, L>() && !quad_late_frame<1, L>() && !quad_late_frame<-LATE_DN, L>())),
                              "which frames are late");   // (16-step lag: frames m-1 and m+1, the only neighbours Q = 2 has)
                constexpr uint32_t kmask = (uint32_t)((MASK >> (R * K1)) & ((1ull << K1) - 1ull));
                load_cells<PA0, -R, L, 0, (quad_late_frame<-R, L>() ? L + 2 : L + 3), kmask, 0, RE>(cx, tu);   // frame m-1 cannot deliver its last half cell yet,
                load_cells<PA0, R, L, 0, (quad_late_frame<R, L>() ? L + 2 : L + 3), kmask, 0, RE>(cx, td);     // nor can frame m+LATE_DN
                rows_sum<Q, L, MASK, PA, R, 0>(a, tu, td, p3A, accA);
                rows_sum<Q, L, MASK, PHB, R, 1>(a, tu, td, p3B, accB);
                rows_sum_ahead<Q, L, MASK, PA + 2, R, 2>(a, tu, td, p3C, qc.accA, qc);
                rows_sum_ahead<Q, L, MASK, PA + 3, R, 3>(a, tu, td, p3D, qc.accB, qc);
                if constexpr (i == 0) LWS_SETPRIO(0);
                if constexpr (i == Q - 2) LWS_SETPRIO(2);
            }
        });
    } else {
        // the sums came with the previous pair; one half cell of two frames was not there yet: finish the group that needs it
        accA = cadd(accA, qc.accA);
        accB = cadd(accB, qc.accB);
        if constexpr (quad_late_frame<-1, L>()) {
            float2 u1[2 * L + 6], d3[2 * L + 6];
            load_cells<PA0, -1, L, L + 2, 1, (uint32_t)((MASK >> K1) & ((1ull << K1) - 1ull)), 0, RE>(cx, u1);
            if constexpr (Q > LATE_DN) load_cells<PA0, LATE_DN, L, L + 2, 1, (uint32_t)((MASK >> ((Q > LATE_DN ? LATE_DN : 0) * K1)) & ((1ull << K1) - 1ull)), 0, RE>(cx, d3);
            if constexpr (r13) {
                quad_finish<Q, L, MASK, PHB>(a, qc, u1[2 * L + 4], d3[2 * L + 4], accB);
            } else {
                if constexpr (LATE_DN == 1) {   // (16-step lag: one group, both of its late taps)
                    quad_finish_plain<Q, L, MASK, PHB, 1>(a, qc, u1[2 * L + 4], accB, d3[2 * L + 4]);
                } else {
                    quad_finish_plain<Q, L, MASK, PHB, 1>(a, qc, u1[2 * L + 4], accB);
                    if constexpr (Q > LATE_DN) quad_finish_plain<Q, L, MASK, PHB, LATE_DN>(a, qc, d3[2 * L + 4], accB);
                }
            }
        }
        LWS_SETPRIO(2);   // (priority 3 here measured the same within run-to-run noise on one box)
    }
    // ---- first bin
    const float tA = amp_cur[PA];
    const float2 outA = project(accA, tA, cx.live && (tA > cx.thr), cr.o0);
    if constexpr (RE == 0) image_store<L, PA, 1>(cx.img_lo, cx.img_hi, cx.img_both, outA);
    else image_publish<L, PA, PA, 1, RE>(cx.uo, st, en, cx.dummy, outA, cx.is_end1);
    // ---- second bin (its centre taps include the first bin's result)
    centre_sum<L, MASK, PHB, PHB, RE>(a, cx, st, en, cr.o1, cr.o2, outA, accB);
    const float tB = amp_cur[PHB];
    const float2 outB = project(accB, tB, cx.live && (tB > cx.thr), cr.o1);
    // both outputs are one ring cell: nobody reads either before the pair is complete (the second bin took the first one's from
    // its register)
    lds_write128(ring_addr<PA, 0, 0, 1>(cx.ob), outA, outB);
    lds_write128(ring_addr<PA, 0, 0, 1>(cx.obh), outA, outB);   // the halo copy of the first / last HALO lanes; every other lane writes its own entry twice
    if constexpr (RE == 0) image_store<L, PHB, 1>(cx.img_lo, cx.img_hi, cx.img_both, outB);
    else image_publish<L, PHB, PHB, 1, RE>(cx.uo, st, en, cx.dummy, outB, cx.is_end1);
    cr.prev_out = outB;
    cr.o0 = cr.o2;
    cr.o1 = cr.o3;
    cr.o2 = make_float2(o45.x, o45.y);
    cr.o3 = make_float2(o45.z, o45.w);
}

// Q = 8: one pair of bins of a HELPER lane (helper H of its slot): the taps of the neighbour frames row_owner() gives it,
// for the pair the slot's main wave reaches help_ahead(H) steps from now; phases (4,5), (6,7), then (0,1), (2,3) of the lane's
// next block.  Same windows, same order of operations per frame pair as compute_pair; no frame of a helper is late.
template <int Q, int L, uint64_t MASK, int PA, int H, int RE = 0>
__device__ __forceinline__ void helper_pair(const SysArgs &a, const LaneCtx &cx, QuadCarry<L> &qc) {
    constexpr int AHEAD = help_ahead(H);
    constexpr int PH = (PA + AHEAD) & 7, CO = PA + AHEAD - PH;   // phase of the first bin; clock of its block
    constexpr int PH0 = PH & ~2;
    static_assert((AHEAD == 4 || AHEAD == 6) && (CO == 0 || CO == 8) && (PH & 1) == 0, "less than a block ahead, mailbox 4 pairs deep");
    float2 accA = make_float2(0.f, 0.f), accB = make_float2(0.f, 0.f);
    if constexpr ((PH & 2) == 0) {
        R13Partials<L> p3;   // (unused: no shared-weight rows here)
        qc.accA = make_float2(0.f, 0.f);
        qc.accB = make_float2(0.f, 0.f);
        static_for<Q - 1>([&](auto ir) {
            constexpr int R = decltype(ir)::value + 1;
            if constexpr (row_owner(R) == H) {
                constexpr uint32_t kmask = (uint32_t)((MASK >> (R * (L + 1))) & ((1ull << (L + 1)) - 1ull));
                if constexpr (kmask != 0) {   // (a frame pair the plan has)
                static_assert(!quad_late_frame<-R, L>() && !quad_late_frame<R, L>(), "late frames stay with the main wave");
                // two steps ahead of the second pair (rows_sum_ahead) and AHEAD ahead of the main wave: the newest tap
                // fetched (fourth bin, +L) must have been produced before this pair started
                static_assert(SKEW * R - L - 3 - AHEAD >= 1 && LAG - SKEW * R - L - 3 - AHEAD >= 1, "helper runs too far ahead");
                float2 tu[2 * L + 6], td[2 * L + 6];
                load_cells<PH0, -R, L, 0, L + 3, kmask, CO, RE>(cx, tu);
                load_cells<PH0, R, L, 0, L + 3, kmask, CO, RE>(cx, td);
                if constexpr (TW) {          // (the bins belong to this block or -- CO = 8 -- to the lane's next one)
                    float2 tau[4];
                    tw_fetch<PH, R>(CO ? cx.tw_nxt : cx.tw, tau);
                    tw_rows<Q, L, MASK, PH, R, 0>(a, tau[0], tu, td, accA, qc);
                    tw_rows<Q, L, MASK, PH + 1, R, 1>(a, tau[1], tu, td, accB, qc);
                    tw_rows<Q, L, MASK, PH + 2, R, 2>(a, tau[2], tu, td, qc.accA, qc);
                    tw_rows<Q, L, MASK, PH + 3, R, 3>(a, tau[3], tu, td, qc.accB, qc);
                } else {
                rows_sum<Q, L, MASK, PH, R, 0>(a, tu, td, p3, accA);
                rows_sum<Q, L, MASK, PH + 1, R, 1>(a, tu, td, p3, accB);
                rows_sum_ahead<Q, L, MASK, PH + 2, R, 2>(a, tu, td, p3, qc.accA, qc);
                rows_sum_ahead<Q, L, MASK, PH + 3, R, 3>(a, tu, td, p3, qc.accB, qc);
                }
                }
            }
        });
    } else {
        accA = qc.accA;
        accB = qc.accB;
    }
    lds_write128(cx.mbox + mbox_addr(0, H, (PA + AHEAD) >> 1), accA, accB);
}

// weight W[0][r][k] (x = r (L+1) + k) for the Nyquist lanes: bin F-1 is a multiple of Q, every twiddle is 1
__device__ __forceinline__ wp_t nyq_weight(const SysArgs &a, int x) {
#if LWS_Q8
    using lds_u64 = const volatile __attribute__((address_space(3))) unsigned long long;
    return *(lds_u64 *)(unsigned)(WNYQ_OFF + x * 8);   // (scaled copy made at kernel start)
#else
    return a.w[x];
#endif
}

// State of the service duties (HBM loader for set 0 and the Nyquist bins of every sweep slot).
struct ServiceState {      // (both as the loads delivered them: raw bits of the storage format)
    float nyq_amp_next;     // Nyquist lanes: target magnitude of the next block's Nyquist bin
    float2 nyq_in_next;     // Nyquist loader lane: previous-sweep Nyquist value of the next block's frame
    float2 nyq_acc;         // Nyquist lanes, two-part call: the neighbour frames' taps, summed one pair ahead
};

// Lane l < NSLOTS computes the Nyquist bin (bin C = F-1) of the frame of sweep slot l whose 512-step period ended at
// clock t0 (phase 0 of the current block); lane NSLOTS feeds set 0 with the stored Nyquist value of that frame.
// Called at the start of the first pair of the block, when every slot has published bins C-1, C-2, ...
// RE = C mod 8 != 0: t0 is still phase 0 of the block, the frames end at its phase RE and the call comes in the pair that
// starts there; bin C is then a multiple of Q only if RE is, hence the quarter turns of the weights below.
// PART: 0 = everything.  When the frames end in the second pair of a quad of bins (RE = 2, 6) that pair is a light one for
// the sweep slots and they would wait for the Nyquist lanes' ~26 dependent weights in it: the neighbour frames' taps (all at
// least 8 steps old) are then summed one pair earlier, where the slots are busy with the quad's tap windows (PART 1), and the
// call in the frame-end pair only adds the frame's own last bins and re-projects (PART 2).
template <int Q, int L, uint64_t MASK, bool MULTI, bool H16, int RE = 0, int PART = 0>
__device__ __forceinline__ void service_nyquist(const SysArgs &a, ServiceState &sv, int lane, int t0, int wg, int n_eff,
                                                int n_groups, const float *thr_eff, void *state_nyq_b,
                                                const void *amp_nyq_b) {
    constexpr int K1 = L + 1;
    const int C = a.C, Kr = a.Kr;
    const int ablk = (t0 >> 3);
    const int slot = lane;                       // lanes 0..NSLOTS-1
    const bool is_nyq_lane = lane < NSLOTS;
    const bool is_nyq_loader = lane == NSLOTS;
    const int v0 = t0 - (is_nyq_lane ? (slot + 1) * LAG : 0);
    // a frame ends every SKEW clocks; with a skew of two blocks only every other block has one (the same blocks for
    // every slot and for the loader: LAG and C are multiples of SKEW)
    if (((v0 + RE - C) & (SKEW - 1)) != 0) return;
    const int vrow = (v0 + RE - C) / SKEW;            // virtual frame whose Nyquist bin is due now (floor: SKEW | v0 - C)
    const int rho = vrow & (ROWL - 1), kap = vrow >> ROWL_SHIFT;
    const int gl = kap / Kr, k = kap - gl * Kr;
    const int g = MULTI ? gl * a.nwg + wg : gl;   // global pass (this workgroup runs passes wg, wg + nwg, ...)
    const int me = k * ROWL + rho;
    const int j = g * NSLOTS + (is_nyq_lane ? slot : -1);
    const bool valid = (v0 + RE - C >= 0) && (me < a.Tp) && (is_nyq_lane ? (j < n_eff) : (is_nyq_loader && g < n_groups));
    if (is_nyq_loader && PART != 1) {
        const float2 nin = raw_value<H16>(sv.nyq_in_next);   // loaded one block ago for this frame
        lds_write(NYQ_OFF + rho * 8, nin);
        lds_write(blk_mod(ablk) * BLK_BYTES + PLR * LANE_B + (RE >> 1) * PAIR_BYTES, nin);   // and as entry "bin C" of set 0's image lane
        const int vr1 = vrow + 1, rho1 = vr1 & (ROWL - 1), kap1 = vr1 >> ROWL_SHIFT;
        const int g1 = kap1 / Kr, k1 = kap1 - g1 * Kr, me1 = k1 * ROWL + rho1;
        if (vr1 >= 0 && me1 < a.Tp) sv.nyq_in_next = load_l2<H16>(state_nyq_b, me1);
    }
    if (is_nyq_lane) {
        const float target = raw_real<H16>(sv.nyq_amp_next);
        const bool real_row = valid && (me >= a.Qa - 1) && (me < a.T + a.Qa - 1);
        const float thr = thr_eff[valid ? j : 0];
        const int set_new = (slot + 1) * SET_BYTES, set_old = slot * SET_BYTES;
        int nb[Q][NBLK], ob[Q][NBLK], nn[Q], no[Q];
#pragma unroll
        for (int d = 0; d < Q; ++d) {
            const int ln = ((rho - d) & (ROWL - 1)), lo = ((rho + d) & (ROWL - 1));
            nn[d] = NYQ_OFF + (slot + 1) * SLOT_BYTES + ln * 8;
            no[d] = NYQ_OFF + slot * SLOT_BYTES + lo * 8;
#pragma unroll
            for (int m = 0; m < NBLK; ++m) {
                const int blk = blk_mod(ablk - m) * BLK_BYTES;
                nb[d][m] = set_new + blk + ln * LANE_B;   // ring_addr adds the HALO offset
                ob[d][m] = set_old + blk + lo * LANE_B;
            }
        }
        const float2 old = lds_read(no[0]);
        float2 acc = PART == 2 ? sv.nyq_acc : make_float2(0.f, 0.f);
        // bin C: bin % Q == 0, every twiddle is 1; taps above Nyquist are conjugated images
        if constexpr (PART != 1) static_for<L>([&](auto ik) {
            constexpr int k = decltype(ik)::value + 1;
            if constexpr ((MASK >> k) & 1ull) {
                const float2 lo = lds_read(ring_addr<RE, -k>(nb[0]));
                pair_rot<0>(acc, nyq_weight(a, k), lo, cj(lo));
            }
        });
        if constexpr (TW && PART != 2) static_for<Q - 1>([&](auto ir) {
            // (LWS_TW) twiddle tau_r(C) of the Nyquist bin: any angle, the same for every lane
            constexpr int r = decltype(ir)::value + 1;
            const float2 tau = wp_value(a.tw_nyq[r]);
            if constexpr ((MASK >> (r * K1)) & 1ull)
                pair_rot<0>(acc, nyq_weight(a, r * K1), cmulf(tau, lds_read(nn[r])), cmulf(cj(tau), lds_read(no[r])));
            static_for<L>([&](auto ik) {
                constexpr int k = decltype(ik)::value + 1;
                if constexpr ((MASK >> (r * K1 + k)) & 1ull) {
                    const float2 up = lds_read(ring_addr<RE, -SKEW * r - k>(nb[r]));
                    const float2 dn = lds_read(ring_addr<RE, SKEW * r - k - LAG>(ob[r]));
                    // W0 tau up + conj(W0 tau) dn + W0 conj(tau) conj(dn) + conj(W0 conj(tau)) conj(up) = W0 X + conj(W0) conj(X)
                    const float2 x = cadd(cmulf(tau, up), cj(cmulf(tau, dn)));
                    pair_rot<0>(acc, nyq_weight(a, r * K1 + k), x, cj(x));
                }
            });
        });
        if constexpr (!TW && PART != 2) static_for<Q - 1>([&](auto ir) {
            constexpr int r = decltype(ir)::value + 1;
            constexpr int rot = eighths<Q>(RE % Q, r) >> 1;    // twiddle exp(2j pi (C mod Q) r / Q) = j^rot, rot even (RE is)
            static_assert((eighths<Q>(RE % Q, r) & 3) == 0, "half turns only");
            if constexpr ((MASK >> (r * K1)) & 1ull)
                pair_rot<rot>(acc, nyq_weight(a, r * K1), lds_read(nn[r]), lds_read(no[r]));
            static_for<L>([&](auto ik) {
                constexpr int k = decltype(ik)::value + 1;
                if constexpr ((MASK >> (r * K1 + k)) & 1ull) {
                    const float2 up = lds_read(ring_addr<RE, -SKEW * r - k>(nb[r]));
                    const float2 dn = lds_read(ring_addr<RE, SKEW * r - k - LAG>(ob[r]));
                    const float2 bsum = make_float2(up.x + dn.x, up.y - dn.y);   // up + conj(dn)
                    const float2 csum = make_float2(dn.x + up.x, dn.y - up.y);   // dn + conj(up)
                    pair_rot<rot>(acc, nyq_weight(a, r * K1 + k), bsum, csum);
                }
            });
        });
        if constexpr (PART == 1) {
            sv.nyq_acc = acc;
            return;
        }
        const bool active = real_row && (target > thr);
        const float2 out = project(acc, target, active, old);
        lds_write(nn[0], out);
        lds_write(set_new + blk_mod(ablk) * BLK_BYTES + PLR * LANE_B + (RE >> 1) * PAIR_BYTES, out);   // bin C of the image lane: production time = this clock
        // (the last slot stores whatever reaches it: idle slots of the last group pass the final values on)
        if ((slot == NSLOTS - 1) && (v0 + RE - C >= 0) && (me < a.Tp) && (g < n_groups)) store_l2<H16>(state_nyq_b, me, out, MULTI);
        // target magnitude of the next block's Nyquist bin
        const int vr1 = vrow + 1, rho1 = vr1 & (ROWL - 1), kap1 = vr1 >> ROWL_SHIFT;
        const int g1 = kap1 / Kr, k1 = kap1 - g1 * Kr, me1 = k1 * ROWL + rho1;
        sv.nyq_amp_next = (vr1 >= 0 && me1 < a.Tp) ? load_real_raw<H16>(amp_nyq_b, me1) : 0.f;   // (raw zero bits = 0 in both formats)
    }
}

#if LWS_Q8
// The same with one lane per (sweep slot, frame offset r): 48 weights in one lane's dependent chain cost the service wave as
// much as a helper wave's frame pair, in the very pair in which the helpers sum.  Lane slot*Q + r sums the taps of frames
// m-+r (r = 0: the frame's own bins C-k and their images), the Q partial sums are combined across the lanes, lane r = 0
// re-projects and stores.  (Weights from the LDS table: the row is a per-lane index.)
// RE = C mod 8 != 0 (frames that end inside a block, see service_nyquist): the call comes in the pair that starts at phase RE,
// and bin C is then not a multiple of Q: lane r's weights carry the twiddle exp(2j pi RE r / 8) = j^(RE r / 2) -- whole quarter
// turns, RE being even -- applied to each weight as it is fetched (the frame offset, hence the turn, is a per-lane value here).
template <int Q, int L, bool MULTI, bool H16, int RE = 0>
__device__ __forceinline__ void service_nyquist_rows(const SysArgs &a, ServiceState &sv, int lane, int t0, int wg, int n_eff,
                                                     int n_groups, const float *thr_eff, void *state_nyq_b,
                                                     const void *amp_nyq_b) {
    static_assert(Q == 8 && NSLOTS * Q < LANES, "one lane per (slot, frame offset), then the loader lane");
    constexpr int K1 = L + 1;
    const int C = a.C, Kr = a.Kr;
    const int ablk = (t0 >> 3);
    const int slot = lane / Q, r = lane - slot * Q;
    const bool is_nyq_lane = lane < NSLOTS * Q;
    const bool is_nyq_loader = lane == NSLOTS * Q;
    const int v0 = t0 - (is_nyq_lane ? (slot + 1) * LAG : 0);
    const int vrow = (v0 + RE - C) / SKEW;       // virtual frame whose Nyquist bin is due now (SKEW | v0 + RE - C)
    const int rho = vrow & (ROWL - 1), kap = vrow >> ROWL_SHIFT;
    const int gl = kap / Kr, k_ = kap - gl * Kr;
    const int g = MULTI ? gl * a.nwg + wg : gl;
    const int me = k_ * ROWL + rho;
    const int j = g * NSLOTS + (is_nyq_lane ? slot : -1);
    const bool valid = (v0 + RE - C >= 0) && (me < a.Tp) && (is_nyq_lane ? (j < n_eff) : (is_nyq_loader && g < n_groups));
    if (is_nyq_loader) {
        const float2 nin = raw_value<H16>(sv.nyq_in_next);
        lds_write(NYQ_OFF + rho * 8, nin);
        lds_write(blk_mod(ablk) * BLK_BYTES + PLR * LANE_B + (RE >> 1) * PAIR_BYTES, nin);
        const int vr1 = vrow + 1, rho1 = vr1 & (ROWL - 1), kap1 = vr1 >> ROWL_SHIFT;
        const int g1 = kap1 / Kr, k1 = kap1 - g1 * Kr, me1 = k1 * ROWL + rho1;
        if (vr1 >= 0 && me1 < a.Tp) sv.nyq_in_next = load_l2<H16>(state_nyq_b, me1);
    }
    if (is_nyq_lane) {
        const float target = raw_real<H16>(sv.nyq_amp_next);
        const bool real_row = valid && (me >= a.Qa - 1) && (me < a.T + a.Qa - 1);
        const float thr = thr_eff[valid ? j : 0];
        const int set_new = (slot + 1) * SET_BYTES, set_old = slot * SET_BYTES;
        const int ln = (rho - r) & (ROWL - 1), lo = (rho + r) & (ROWL - 1);
        // the taps at times -SKEW r - k (frame m-r, this sweep) and SKEW r - k - LAG (frame m+r, previous sweep): r whole ring
        // blocks before / after the block of time -k
        const int bn = set_new + blk_mod(ablk - r - 1) * BLK_BYTES + (ln + HALO) * LANE_B;
        const int bo = set_old + blk_mod(ablk + r - 1) * BLK_BYTES + (lo + HALO) * LANE_B;
        // (RE != 0: the taps k <= RE lie in the block of the frame end itself, one ring block later)
        const int bn1 = set_new + blk_mod(ablk - r) * BLK_BYTES + (ln + HALO) * LANE_B;
        const int bo1 = set_old + blk_mod(ablk + r) * BLK_BYTES + (lo + HALO) * LANE_B;
        // (LWS_TW: the lane's twiddle tau_r(C) is any complex number: taps turned as in service_nyquist, weights as they are)
        const float2 tau = TW ? lds_read(TWNYQ_OFF + (r & 7) * 8) : make_float2(1.f, 0.f);
        const int rot = (RE && !TW) ? ((RE * r) >> 1) & 3 : 0;        // quarter turns of this lane's weights
        const float sg = (rot & 1) ? -1.f : 1.f;             // b = um +- dp, c = dm +- up: minus for an odd quarter turn
        auto turned = [&](wp_t w) -> wp_t {
            if constexpr (RE == 0 || TW) return w;
            float wr = __uint_as_float((unsigned)(w & 0xffffffffull)), wi = __uint_as_float((unsigned)(w >> 32));
            const float t = wr;
            wr = (rot & 1) ? -wi : wr; wi = (rot & 1) ? t : wi;
            wr = (rot & 2) ? -wr : wr; wi = (rot & 2) ? -wi : wi;
            return ((unsigned long long)__float_as_uint(wi) << 32) | __float_as_uint(wr);
        };
        const int nn = NYQ_OFF + (slot + 1) * SLOT_BYTES + ln * 8, no = NYQ_OFF + slot * SLOT_BYTES + lo * 8;
        const bool centre = r == 0;                 // its "frame m+r" terms are the images: dn = 0 below gives b = up, c = conj(up)
        // (lws::tw_q8: a plan with Q < 8 has no frames m-+r for r >= Q.  Their weights are zero, but what sits at those ring
        //  positions near the ends of the spectrogram was never written -- and zero times a stale NaN is not zero)
        const bool dead = TW && r >= a.Qa;
        const float2 zero = make_float2(0.f, 0.f);
        float2 acc = zero;
        {
            float2 un = lds_read(nn), dn = lds_read(no);
            if constexpr (TW) { un = cmulf(tau, un); dn = cmulf(cj(tau), dn); }
            pair_rot<0>(acc, turned(nyq_weight(a, r * K1)), (centre || dead) ? zero : un, (centre || dead) ? zero : dn);
        }
        static_for<L>([&](auto ik) {
            constexpr int k = decltype(ik)::value + 1, within = (RE - k) & 7;   // time RE - k of the block, or of the one before
            constexpr bool same_block = RE - k >= 0;
            constexpr int off = (within >> 1) * PAIR_BYTES + (within & 1) * 8;
            float2 up = lds_read((same_block ? bn1 : bn) + off);
            float2 dn = lds_read((same_block ? bo1 : bo) + off);
            dn = (centre || dead) ? zero : dn;
            up = dead ? zero : up;
            float2 bsum, csum;
            if constexpr (TW) {                                 // W0 X + conj(W0) conj(X), X = tau up + conj(tau dn)  (service_nyquist)
                bsum = cadd(cmulf(tau, up), cj(cmulf(tau, dn)));
                csum = cj(bsum);
            } else if constexpr (RE == 0) {
                bsum = make_float2(up.x + dn.x, up.y - dn.y);   // up + conj(dn)
                csum = make_float2(dn.x + up.x, dn.y - up.y);   // dn + conj(up)
            } else {
                bsum = make_float2(up.x + sg * dn.x, up.y - sg * dn.y);   // up +- conj(dn)
                csum = make_float2(dn.x + sg * up.x, dn.y - sg * up.y);   // dn +- conj(up)
            }
            pair_rot<0>(acc, turned(nyq_weight(a, r * K1 + k)), bsum, csum);
        });
        auto dpp = [](float x, auto ctrl) {
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
        };
        auto sum8 = [&](float v) {
            v += dpp(v, std::integral_constant<int, 0xB1>());    // quad_perm [1,0,3,2]
            v += dpp(v, std::integral_constant<int, 0x4E>());    // quad_perm [2,3,0,1]
            v += dpp(v, std::integral_constant<int, 0x141>());   // row_half_mirror
            return v;
        };
        acc.x = sum8(acc.x);
        acc.y = sum8(acc.y);
        const float2 old = lds_read(no);            // (lane r = 0: the frame's own previous value)
        const bool active = real_row && (target > thr);
        const float2 out = project(acc, target, active, old);
        if (centre) {
            lds_write(nn, out);
            lds_write(set_new + blk_mod(ablk) * BLK_BYTES + PLR * LANE_B + (RE >> 1) * PAIR_BYTES, out);
            if ((slot == NSLOTS - 1) && (v0 + RE - C >= 0) && (me < a.Tp) && (g < n_groups)) store_l2<H16>(state_nyq_b, me, out, MULTI);
        }
        const int vr1 = vrow + 1, rho1 = vr1 & (ROWL - 1), kap1 = vr1 >> ROWL_SHIFT;
        const int g1 = kap1 / Kr, k1 = kap1 - g1 * Kr, me1 = k1 * ROWL + rho1;
        sv.nyq_amp_next = (vr1 >= 0 && me1 < a.Tp) ? load_real_raw<H16>(amp_nyq_b, me1) : 0.f;
    }
}
#endif

// MULTI: several workgroups share a spectrogram (a.nwg > 1); the single-workgroup instantiation carries none of it
// RE = (F-1) mod 8 (see th0)
template <int Q, int L, uint64_t MASK, bool MULTI, bool H16, int RE = 0>
__global__ void __launch_bounds__(NTHREADS, (NTHREADS + 255) / 256) k_systolic(SysArgs a_in) {
    static_assert((RE & 1) == 0 && RE >= 0 && RE < 8, "frame end phase");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (a_in.gate != nullptr && __hip_atomic_load(a_in.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    const int nwg = MULTI ? a_in.nwg : 1;
    const int b = MULTI ? blockIdx.x / nwg : blockIdx.x, wg = MULTI ? blockIdx.x - b * nwg : 0;
    // The weighted sums are only ever normalised (project()), so the weights of this spectrogram are scaled by the power
    // of two that brings its largest magnitude to [1, 2): exact, and |acc|^2 can then be formed in fp32 without the
    // rescue path for very small (or large) data that project() would otherwise need for every bin.
    // (With fp16 storage the data itself arrives multiplied by that power of two -- see Store -- so the weights stay as
    // they are and the thresholds are scaled instead.)
    SysArgs a = a_in;
    const float data_scale = store_scale(a_in.amax[b]);
    {
        const float sc = H16 ? 1.0f : data_scale;
#if LWS_Q8
        // (hardware wave -> role table below) the list of the wave's role: mains 0, helpers 1 / 2
        const int hw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#if LWS_TWQ
        const int owner = hw >= 4 ? 1 : 0;            // (role = hardware wave: mains 0..2, service 3, the helper of slot s 4 + s)
#else
        const int owner = (hw == 2 || hw == 3) ? 1 : ((hw == 4 || hw == 5) ? 2 : 0);
#endif
#pragma unroll
        for (int x = 0; x < WLIST; ++x) {
            const unsigned long long u = owner == 0 ? a_in.w[x] : (owner == 1 ? a_in.w[WLIST + x] : a_in.w[2 * WLIST + x]);
            const float re = __uint_as_float((unsigned)(u & 0xffffffffull)) * sc;
            const float im = __uint_as_float((unsigned)(u >> 32)) * sc;
            a.w[x] = ((unsigned long long)__float_as_uint(im) << 32) | __float_as_uint(re);
        }
        if constexpr (TW) {             // ... and their twiddles (static indices: a run-time index would put the arguments in scratch)
            static_for<8>([&](auto ir) {
                constexpr int r = decltype(ir)::value;
                if (threadIdx.x == 64 + r) reinterpret_cast<float2 *>(smem + TWNYQ_OFF)[r] = wp_value(a_in.tw_nyq[r]);
            });
        }
        if (threadIdx.x < QMAX * 6) {   // the Nyquist lanes' table (the __syncthreads() below publishes it)
            const unsigned long long u = a_in.w[WNYQ + threadIdx.x];
            const float re = __uint_as_float((unsigned)(u & 0xffffffffull)) * sc;
            const float im = __uint_as_float((unsigned)(u >> 32)) * sc;
            reinterpret_cast<float2 *>(smem + WNYQ_OFF)[threadIdx.x] = make_float2(re, im);
        }
#else
#pragma unroll
        for (int x = 0; x < NW; ++x) {
            const float re = __uint_as_float((unsigned)(a_in.w[x] & 0xffffffffull)) * sc;
            const float im = __uint_as_float((unsigned)(a_in.w[x] >> 32)) * sc;
            a.w[x] = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(im)) << 32) |
                     (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(re));
        }
#endif
    }
    // `wave` is the ROLE of a wave (sweep slot 0..NSLOTS-1, or NSLOTS = service), not its hardware index: hardware
    // waves w and w + 4 share a SIMD and the older one is served first.  Every sweep slot meets the service wave at every
    // pair of bins (it feeds slot 0, computes the Nyquist bins and writes the results back), so the service wave takes
    // hardware wave 3 and is never kept waiting for issue slots; the slot that gives up that place runs on hardware wave 7,
    // beside it, where the service wave's light instruction stream leaves most of the SIMD free.
#ifndef LWS_ROLE_SWAP
#define LWS_ROLE_SWAP (!LWS_WIDE)
#endif
    // Wide build: a sweep slot is two waves (the lower and the upper 64 lanes of the 128-lane ring row: role = 2 slot + half)
    // and there are two service waves, one per half (roles 6 and 7: each loads and writes back its 64 lanes; the first one
    // also computes the Nyquist bins).
    const int hw_wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // provably uniform
#if LWS_Q8
    // hardware waves w and w + 4 share a SIMD: main 0 | main 1 | helper 1 of slot 0 | helper 1 of slot 1 on SIMDs 0..3, then the
    // (lighter) second helpers beside the mains and the service wave beside a first helper
#if LWS_TWQ
    // three slots of a main and one helper wave: main s on SIMD s with its helper beside it, the service wave alone on the fourth
    static_assert(NSLOTS == 3 && NHELP == 1 && WPS == 1 && NWAVES == 7, "role table");
    const int wave = hw_wave;
#else
    static_assert(NSLOTS == 2 && NHELP == 2 && WPS == 1, "role table");   // (the weight lists above follow the same table)
    const int wave = hw_wave < 2 ? hw_wave : (hw_wave == 2 ? 3 : (hw_wave == 3 ? 5 : (hw_wave == 4 ? 4 : (hw_wave == 5 ? 6 : 2))));
#endif
#else
#ifndef LWS_WIDE_ROLEMAP
#define LWS_WIDE_ROLEMAP LWS_WIDE
#endif
    // Wide build (round 3): hardware waves h and h + 4 -- one SIMD -- carry the two halves of one slot, SIMD 3 both service waves.
    // With role = hardware wave the service waves shared their SIMDs with the halves of slot 1 and got every other issue slot
    // there, and every slot meets them at every pair: 38.0 -> 40.6 % (64 x 6000 x 1025, 60 sweeps).
    const int rm = a_in.rolemap;   // (experiment hook, LWS_SYSTOLIC_ROLEMAP)
    const int wave = (LWS_WIDE_ROLEMAP && WPS == 2 && NWAVES == 8)
                         ? (rm == 1 ? hw_wave
                            : rm == 2 ? ((hw_wave & 3) == 3 ? 6 + (hw_wave >> 2) : ((hw_wave & 3) == 2 ? 4 + (hw_wave >> 2) : (hw_wave & 3) + 2 * (hw_wave >> 2)))
                            : rm == 3 ? ((hw_wave & 3) == 3 ? 6 + (hw_wave >> 2) : (hw_wave & 3) * 2 + 1 - (hw_wave >> 2))
                            : ((hw_wave & 3) * 2 + (hw_wave >> 2)))
                   : (LWS_R16 && NWAVES == 16 && rm != 1) ? (hw_wave == 3 ? 15 : (hw_wave == 15 ? 3 : hw_wave))
                   : ((LWS_ROLE_SWAP && NCOMPUTE == 7 && WPS == 1 && rm != 1) ? (rm == 2 ? (hw_wave == 0 ? 7 : (hw_wave == 7 ? 0 : hw_wave)) : (hw_wave == 3 ? 7 : (hw_wave == 7 ? 3 : hw_wave))) : hw_wave);
#endif
    const int hf = wave % WPS;                     // which half of the ring row this wave's lanes are
    const int sub = SPW > 1 ? lane / ROWL : 0;     // short-frame builds: which of the wave's sweep slots the lane belongs to
    const int rl = SPW > 1 ? lane % ROWL : hf * LANES + lane;   // the lane's place in the ring row = its frame within a round
    float *thr_eff = reinterpret_cast<float *>(smem + THR_OFF);
    int *meta = reinterpret_cast<int *>(smem + META_OFF);
    const int G = a.G, C = a.C, Kr = a.Kr;
    using ST = Store<H16>;
    void *state_w_b = static_cast<char *>(a.state_w) + (size_t)b * G * ROWL * ST::CB;
    const void *amp_w_b = static_cast<const char *>(a.amp_w) + (size_t)b * G * ROWL * ST::RB;
    void *state_nyq_b = static_cast<char *>(a.state_nyq) + (size_t)b * a.TpPad * ST::CB;
    const void *amp_nyq_b = static_cast<const char *>(a.amp_nyq) + (size_t)b * a.TpPad * ST::RB;
    constexpr int T_START = -8;   // one block of warm-up

    // sweeps whose threshold is not below the largest magnitude cannot change anything: drop them
    if (threadIdx.x == 0) {
        const float amax = a.amax[b];
        int n = 0;
        for (int i = 0; i < a.n_iters; ++i) {
            const float th = a.thr[(size_t)b * a.n_iters + i];
            if (amax > th) thr_eff[n++] = H16 ? th * data_scale : th;
        }
        meta[0] = n;
    }
    // poison-free start: rings may hold anything, but zero keeps the arithmetic of idle lanes finite
    for (int i = threadIdx.x; i < THR_OFF / 8; i += NTHREADS) reinterpret_cast<float2 *>(smem)[i] = make_float2(0.f, 0.f);
    for (int i = threadIdx.x; i < MBOX_BYTES / 8; i += NTHREADS) reinterpret_cast<float2 *>(smem + MBOX_OFF)[i] = make_float2(0.f, 0.f);
    if (threadIdx.x < 16) reinterpret_cast<int *>(smem + DONE_OFF)[threadIdx.x] = T_START;
    if constexpr (TW)   // the twiddle table: (P + 8) rows of four float2
        for (int i = threadIdx.x; i < (a_in.tw_P + 8) * (TW_ROW / 8); i += NTHREADS)
            reinterpret_cast<float2 *>(smem + TW_OFF)[i] = reinterpret_cast<const float2 *>(a_in.tw_table)[i];
    __syncthreads();
    const int n_eff = meta[0];
    if (n_eff == 0) return;
    const int n_groups = (n_eff + NSLOTS - 1) / NSLOTS;
    // Passes (groups of NSLOTS sweeps) are dealt round-robin to the nwg workgroups of the spectrogram: this one runs the
    // global passes wg, wg + nwg, ... back to back on its own clock ("local pass" gl = clock / G), each one trailing the
    // pass before it -- run by the previous workgroup of the ring -- by a few hundred steps through HBM (see below).
    const int n_local = (wg < n_groups) ? (n_groups - wg + nwg - 1) / nwg : 0;
    if (n_local == 0) return;
    // slot i runs on clock v_i = t - (i+1)*LAG; the loader (virtual slot -1) on clock t
    // (the + 8 + ...: the last rows a consumer workgroup waits for -- need_max below, frames of ROWP bins -- must be covered by
    //  the progress this workgroup publishes before it leaves the loop: LAG + ROWP - 8 >= C + 16 needs 16 more steps at a 16-step lag)
    const int t_end = (n_local - 1) * G + (NSLOTS + 1) * LAG + SKEW * a.Tp + ROWP + 8 + (LAG < 32 ? 32 - LAG : 0);
    // ---- hand-over between the workgroups of a spectrogram (nwg > 1).  Row r of the skewed state written by pass g is
    // read by pass g + 1.  A workgroup publishes (release, agent scope) the number of rows -- in its own clock -- its
    // last slot and its Nyquist lanes have completed; the loader of the next workgroup in the ring waits (acquire)
    // until the rows it is about to fetch are there.  The producer of local pass gl is the previous workgroup's local
    // pass gl, or for workgroup 0 the last workgroup's local pass gl - 1 (one pass = G rows earlier on its clock).
    // (one counter per workgroup and half of the row: a service wave hands over, and waits for, its own 64 lanes)
    const unsigned *prod_progress = a.progress + ((size_t)b * nwg + (wg ? wg - 1 : nwg - 1)) * WPS + hf;
    unsigned *my_progress = a.progress + ((size_t)b * nwg + wg) * WPS + hf;
    const int prod_shift = wg ? 0 : G;
    // Last row, on the producer's clock, that any valid frame of my last pass reads (frame me starts SKEW*me rows into
    // a pass, so lanes are up to SKEW*63 rows apart and a pass ends later for the later frames): rows beyond it are
    // fetched but never used, and the producer's counter stops short of them.  For workgroup 0 the producer's pass is
    // one local pass behind; its first pass reads the caller's data and waits for nobody.
    const int prod_passes = n_local - (wg ? 0 : 1);   // passes of the producer this workgroup consumes
    const int need_max = prod_passes > 0 ? (prod_passes - 1) * G + SKEW * a.Tp + C + 16 : 0;
    bool gave_up = false;                        // the hand-over failed somewhere in this launch: stop waiting, the results
                                                 // are discarded and the call is re-run with one workgroup per spectrogram
    auto wait_rows = [&](int need) {            // service wave: rows < need of my clock must be complete
        need -= prod_shift;
        if (need > need_max) need = need_max;
        if (need <= 0 || gave_up) return;
        int spins = 0;
        // data rows are written through (store_l2) and read past the L1 and the XCD's L2 (load_l2), so no cache-wide
        // write-back / invalidate is needed: a counter that is only raised after the producer's stores have completed
        while ((int)__hip_atomic_load(prod_progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            __builtin_amdgcn_s_sleep(8);
            ++spins;
            // a co-scheduling failure (workgroups of one ring not resident together), not a hang: flag it for everybody
            if (spins > a.spin_limit) { if (lane == 0) __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); gave_up = true; break; }
            if ((spins & 1023) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { gave_up = true; break; }
        }
        asm volatile("" ::: "memory");
    };

    const bool is_compute = wave < NCOMPUTE;
    const bool is_helper = NHELP > 0 && wave >= NCOMPUTE + WPS;
    const bool is_service = !is_compute && !is_helper;
    [[maybe_unused]] const int helper_no = is_helper ? (wave - NCOMPUTE - WPS) % (NHELP > 0 ? NHELP : 1) + 1 : 0;   // 1 .. NHELP
    // slot0: the wave's (first) sweep slot, wave-uniform; slot: the lane's (SPW > 1: slot0 + sub for a compute wave)
    const int slot0 = is_compute ? (wave / WPS) * SPW : (is_helper ? (wave - NCOMPUTE - WPS) / (NHELP > 0 ? NHELP : 1) : NSLOTS);
    const int slot = (SPW > 1 && is_compute) ? slot0 + sub : slot0;
    LaneCtx cx;
    Carry cr;
    cr.o0 = cr.o1 = cr.o2 = cr.o3 = cr.prev_out = make_float2(0.f, 0.f);
    // target magnitudes of the current block's 8 bins and (in flight) of the next block's: all 8 global loads of a
    // block are issued together one block ahead, so no step ever waits on HBM latency
    // (a wave is either a sweep slot or the service wave: the 16 registers below hold amp_cur[8] | amp_nxt[8] for the
    //  former and the loader's 8 complex values in flight for the latter)
    float amp_cur[8], amp_nxt[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) amp_cur[i] = amp_nxt[i] = 0.f;
    QuadCarry<L> qc;   // sweep slots: what the first pair of a quad hands to the second
    bool wb_cur = false, wb_prev = false;   // service wave: does the last slot's lane have a bin to write back in this / the previous block
    ServiceState sv;
    sv.nyq_amp_next = 0.f;
    sv.nyq_in_next = make_float2(0.f, 0.f);
    sv.nyq_acc = make_float2(0.f, 0.f);
    if (is_service) {
        if constexpr (MULTI) wait_rows(8 + 40);
#pragma unroll
        for (int i = 0; i < 8; ++i) {   // clocks 0..7
            const float2 v = load_l2<H16>(state_w_b, (size_t)i * ROWL + rl);
            amp_cur[i] = v.x;
            amp_nxt[i] = v.y;
        }
    }
    const int set_old = slot * SET_BYTES;   // ring set this slot reads "old" values from; it writes the next one
    // flow control: lane l watches wave l.  A compute wave waits for its producer, its consumer and the service wave;
    // the service wave for every compute wave (loader overwrites what slot 0 reads; the Nyquist lanes read every slot).
    // (two waves per slot: also the other half of the own slot -- the lanes next to the middle of the row read each other's
    // output -- and both halves of the neighbouring slots and of the service.  The two service waves wait for each other as
    // well: the Nyquist lanes of the first one read, 25..29 steps back, entries of set 0 that the second one's loader
    // rewrites two pairs later, so neither may run a pair ahead of the other.)
    // (Q = 8: a main wave also waits for the helpers of its slot -- their mailbox entries for this pair were written two pairs
    // ago -- and a helper for the waves whose output it reads: the main waves of its slot and of the previous one, and the
    // service wave.  Everybody is in the same pair at all times; a helper just works on data for two pairs later.)
    bool watched = false;
    if (lane < NWAVES) {
        const bool whelp = lane >= NCOMPUTE + WPS;
        // the (first) slot of wave `lane`, and this wave's: neighbouring waves hold neighbouring slots in every build
        const int wslot = lane < NCOMPUTE ? (lane / WPS) * SPW : (whelp ? (lane - NCOMPUTE - WPS) / (NHELP > 0 ? NHELP : 1) : NSLOTS);
        if (is_service) watched = lane != wave;
        else if (is_helper) watched = !whelp && (wslot == slot0 - 1 || wslot == slot0 || wslot == NSLOTS);
        else watched = (lane != wave) && (whelp ? wslot == slot0 : (wslot == slot0 - SPW || wslot == slot0 || wslot == slot0 + SPW || wslot == NSLOTS));
    }
    // Where a lane is in a block of 8 steps (which frame of which sweep, first / last bins of the frame, active at
    // all): evaluated once per block for the FOLLOWING block and shifted.  The lanes of a wave sit in at most two
    // consecutive rounds of 64 frames ("kap" = clock / frame period, minus one for the lanes whose frame has not come
    // round yet), so everything that depends on the round -- pass, frame index base, sweep number, threshold -- is
    // kept as wave-uniform state for those two rounds and advanced (no division) when the clock enters a new round; a
    // lane only selects between the two.
    const float inv_kr = 1.0f / (float)Kr;   // (service wave: one division per block in floating point, exact for these magnitudes)
    struct Round { int gl, k, meb; bool okj; float thr; };    // wave-uniform: local pass, round within it, first frame, sweep valid, its threshold
    // (one set of this state per sweep slot of the wave: SPW > 1, the short-frame builds)
    Round rcur[SPW], rprev[SPW];
    int ks_state[SPW];                       // round of lane 0 at the last evaluation (negative: not started)
#pragma unroll
    for (int q = 0; q < SPW; ++q) {
        rcur[q].gl = 0; rcur[q].k = 0; rcur[q].meb = 0; rcur[q].okj = false; rcur[q].thr = 0.f;
        rprev[q] = rcur[q];
        ks_state[q] = (T_START - (slot0 + q + 1) * LAG) >> ROWP_SHIFT;
    }
    struct BlockInfo { bool live, start, end, end1; float thr; int tw; };
    auto block_info = [&](int vblock0) {      // vblock0: the clock of the wave's first slot at phase 0 of the block (wave-uniform)
        // the round bookkeeping of each of the wave's slots: wave-uniform, once per round
        static_for<SPW>([&](auto iq) {
            constexpr int q = decltype(iq)::value;
            const int ks = (vblock0 - q * LAG) >> ROWP_SHIFT;      // slot slot0 + q trails slot0 by q lags
            if (ks != ks_state[q]) {                 // once per round: ROWL blocks
                ks_state[q] = ks;
                rprev[q] = rcur[q];
                if (ks <= 0) { rcur[q].gl = 0; rcur[q].k = 0; }
                else if (++rcur[q].k == Kr) { rcur[q].k = 0; ++rcur[q].gl; }
                const int j = (rcur[q].gl * nwg + wg) * NSLOTS + slot0 + q;
                rcur[q].okj = is_compute && ks >= 0 && j < n_eff;
                rcur[q].meb = rcur[q].k * ROWL;
                rcur[q].thr = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(thr_eff[rcur[q].okj ? j : 0])));
            }
        });
        // the lane's own slot (SPW > 1: picked from the wave's SPW sets of round state), then its place in the block
        int meb_c = rcur[0].meb, meb_p = rprev[0].meb;
        bool ok_c = rcur[0].okj, ok_p = rprev[0].okj;
        float thr_c = rcur[0].thr, thr_p = rprev[0].thr;
        static_for<SPW - 1>([&](auto iq) {
            constexpr int q = decltype(iq)::value + 1;
            const bool mine = sub == q;
            meb_c = mine ? rcur[q].meb : meb_c; meb_p = mine ? rprev[q].meb : meb_p;
            ok_c = mine ? rcur[q].okj : ok_c; ok_p = mine ? rprev[q].okj : ok_p;
            thr_c = mine ? rcur[q].thr : thr_c; thr_p = mine ? rprev[q].thr : thr_p;
        });
        const int rem = (vblock0 - (SPW > 1 ? sub * LAG : 0)) & (ROWP - 1);
        const bool here = SKEW * rl <= rem;                      // this lane's frame of the current round has started
        const int cbase = (rem - SKEW * rl) & (ROWP - 1);
        const int me = (here ? meb_c : meb_p) + rl;
        const bool valid = (here ? ok_c : ok_p) && (me < a.Tp);
        BlockInfo bi;
        bi.live = valid && (me >= a.Qa - 1) && (me < a.T + a.Qa - 1) && (cbase < C);
        bi.start = (cbase == 0);
        bi.end = (cbase == C - th0(RE));
        bi.end1 = (RE != 0) && (cbase == C - th1(RE));
        bi.thr = here ? thr_c : thr_p;
        bi.tw = 0;
        if constexpr (TW) {   // the block's first bin mod P (cbase < 4096: the float quotient is off by one at most)
            int r = cbase - (int)((float)cbase * a.tw_invP) * a.tw_P;
            r += (r < 0) ? a.tw_P : 0;
            r -= (r >= a.tw_P) ? a.tw_P : 0;
            bi.tw = TW_OFF + r * TW_ROW;
        }
        return bi;
    };
    int dlo[NDR];   // per-lane constants of the image-cell offsets: (PLL - HALO - DR - lane) * 16
#pragma unroll
    for (int d = 0; d < NDR; ++d) dlo[d] = (PLL - HALO - (d - HALO) - rl) * LANE_B;
    BlockInfo nxt_bi = block_info(T_START - (slot0 + 1) * LAG);
    int vmod = __builtin_amdgcn_readfirstlane((((T_START - (slot0 + 1) * LAG) % G) + G) % G - 8);   // advanced at the loop head
    int tmod = __builtin_amdgcn_readfirstlane(((T_START % G) + G) % G - 8);
    // short-frame builds: the last slot's clock mod G (the rows of the write-back; NSLOTS * LAG can be several G there)
    int wmod = __builtin_amdgcn_readfirstlane((((T_START - NSLOTS * LAG) % G) + G) % G - 8);
    // The step loop, per ROLE of the wave.  Q = 8 instantiates it once per role (main / helper 1 / helper 2 / service), so that
    // a wave only carries the registers of its own duties (the union does not fit 256 VGPRs); the other builds run one loop
    // with wave-uniform branches (ROLE < 0).
    auto role_loop = [&](auto role_c) __attribute__((always_inline)) {
        constexpr int ROLE = decltype(role_c)::value;
        constexpr int R_SERVICE = 9;
        const bool r_compute = ROLE < 0 ? is_compute : ROLE == 0;
        const bool r_service = ROLE < 0 ? is_service : ROLE == R_SERVICE;
        const bool r_helper = ROLE < 0 ? is_helper : (ROLE >= 1 && ROLE < R_SERVICE);
        for (int t0 = T_START; t0 < t_end; t0 += 8) {
            const int v0 = t0 - (slot0 + 1) * LAG;  // clock of the wave's (first) sweep slot at phase 0 of the block (multiple of 8; its
                                                    // other slots trail by whole rings: same ring block, same phase)
            // ---- block prologue: where is this lane in this block and in the next one?
            const int ablk = (v0 >> 3);
            {
                const BlockInfo cur = nxt_bi;
                nxt_bi = block_info(v0 + 8);
                cx.live = cur.live; cx.is_start = cur.start; cx.is_end = cur.end; cx.is_end1 = cur.end1; cx.thr = cur.thr;
                cx.tw = cur.tw; cx.tw_nxt = nxt_bi.tw;
                cx.nxt_live = nxt_bi.live; cx.nxt_start = nxt_bi.start; cx.nxt_end = nxt_bi.end; cx.nxt_end1 = nxt_bi.end1;
                cx.nxt_thr = nxt_bi.thr;
            }
            cx.lane8 = rl * 8;
            cx.nyq_base = NYQ_OFF + (slot + 1) * SLOT_BYTES + rl * 8;
            cx.dummy = DUMMY_OFF + lane * 8;
            cx.mbox = ((r_service ? 0 : slot) * NHELP * 4 * LANES + lane) * 16;
            cx.halo_shift = (rl < HALO) ? ROWL * LANE_B : (rl >= ROWL - HALO ? -ROWL * LANE_B : 0);
    #pragma unroll
            for (int d = 0; d < NDR; ++d) {   // (only the entries of frames that exist, |DR| <= Q-1, are ever read)
                cx.wlo[d] = cx.is_start ? dlo[d] : 0;
                cx.whi[d] = cx.is_end ? dlo[d] + LANE_B : 0;   // PLR = PLL + 1
                if constexpr (RE != 0) cx.whi1[d] = cx.is_end1 ? dlo[d] + LANE_B : 0;
            }
    #pragma unroll
            for (int m = 0; m < NBLK; ++m) {
                const int blk = blk_mod(ablk - m) * BLK_BYTES;
                cx.uo[m] = set_old + blk;
                cx.ob[m] = set_old + blk + rl * LANE_B;
                cx.obh[m] = cx.ob[m] + cx.halo_shift;
            }
            {
                const int nowhere = SCRATCH_OFF + lane * 8 - (SET_BYTES + PLL * LANE_B);
                const int lo_row = cx.uo[ImageBlocks<L>::m_lo], hi_row = cx.uo[ImageBlocks<L>::m_hi] + LANE_B;   // PLR = PLL + 1
                cx.img_lo = cx.is_start ? lo_row : nowhere;
                cx.img_hi = cx.is_end ? hi_row : nowhere;
                cx.img_both = cx.is_start ? lo_row : cx.img_hi;
            }
            vmod += 8; vmod -= (vmod >= G) ? G : 0;   // v0 mod G and t0 mod G (G is a multiple of 8), wave-uniform
            tmod += 8; tmod -= (tmod >= G) ? G : 0;
            if constexpr (SPW > 1) { wmod += 8; wmod -= (wmod >= G) ? G : 0; }
            if (r_compute) {
                int vnext = vmod + 8;
                vnext -= (vnext >= G) ? G : 0;     // G is a multiple of 8: the next block does not wrap inside
    #pragma unroll
                for (int i = 0; i < 8; ++i) amp_cur[i] = raw_real<H16>(amp_nxt[i]);   // (fp16 storage: the conversion takes the place of the move)
                if constexpr (RE != 0) {   // the dead phases of a frame's last block: never above any threshold
    #pragma unroll
                    for (int i = RE; i < 8; ++i) amp_cur[i] = cx.is_end ? -__builtin_inff() : amp_cur[i];
                }
                // The loads are asm statements so that they land in amp_nxt's own registers and nobody waits for them here
                // (written as plain loads the compiler fetches into temporaries and copies -- i.e. waits -- at once: a stall
                // of one memory latency per block).  The wait is explicit, at the end of the block; these are the only
                // vector-memory operations of a sweep slot.
                if constexpr (SPW > 1) {           // the lane's own slot is `sub` lags behind the wave's first
                    vnext -= sub * LAG;
                    vnext += (vnext < 0) ? G : 0;
                }
                const char *ap = static_cast<const char *>(amp_w_b) + ((size_t)vnext * ROWL + rl) * ST::RB;
                // (the instruction's offset field holds 12 bits: rows 4..7 of a 256-lane row go through a second base)
                constexpr bool far_rows = 7 * ROWL * 4 > 4095;
                const char *ap4 = far_rows ? ap + 4 * ROWL * ST::RB : ap;
    #define LWS_AMP_LOAD(i)                                                                                                             \
        do {                                                                                                                            \
            constexpr int io_ = (far_rows && (i) >= 4) ? (i) - 4 : (i);                                                                 \
            const char *ab_ = (far_rows && (i) >= 4) ? ap4 : ap;                                                                        \
            if constexpr (H16) asm volatile("global_load_ushort %0, %1, off offset:%2" : "=v"(amp_nxt[i]) : "v"(ab_), "i"(io_ * ROWL * 2) : "memory"); \
            else asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(amp_nxt[i]) : "v"(ab_), "i"(io_ * ROWL * 4) : "memory"); \
        } while (0)
                LWS_AMP_LOAD(0); LWS_AMP_LOAD(1); LWS_AMP_LOAD(2); LWS_AMP_LOAD(3);
                LWS_AMP_LOAD(4); LWS_AMP_LOAD(5); LWS_AMP_LOAD(6); LWS_AMP_LOAD(7);
    #undef LWS_AMP_LOAD
            }
            // ---- 4 pairs of bins, phases static
            static_for<4>([&](auto ip) {
                constexpr int PA = 2 * decltype(ip)::value;
                flow_wait(lane, t0 + PA, watched);
                if (r_compute) compute_pair<Q, L, MASK, PA, H16, RE>(a, cx, cr, amp_cur, qc);
                if constexpr (NHELP > 0) {
                    if (r_helper) {
                        if constexpr (PA == 8 - help_ahead(ROLE >= 1 && ROLE <= NHELP ? ROLE : 1)) {   // from here on the lane's next block: that block's frame edges
    #pragma unroll
                            for (int d = 0; d < NDR; ++d) {
                                cx.wlo[d] = cx.nxt_start ? dlo[d] : 0;
                                cx.whi[d] = cx.nxt_end ? dlo[d] + LANE_B : 0;
                                if constexpr (RE != 0) cx.whi1[d] = cx.nxt_end1 ? dlo[d] + LANE_B : 0;
                            }
                        }
                        if constexpr (ROLE >= 1 && ROLE <= NHELP) helper_pair<Q, L, MASK, PA, ROLE, RE>(a, cx, qc);
                    }
                }
                if (a.stress != 0 && ((a.stress >> wave) & 1) && PA + 1 == ((a.stress >> 16) & 7)) {   // test hook, see SysArgs (pairs 1, 3, 5, 7)
                    for (int q = 0; q < 10; ++q) __builtin_amdgcn_s_sleep(32);
                }
                if constexpr (PA == 0 && MULTI) {
                    if (r_service) {
                        // every slot has finished the previous block (flow_wait above); this wave has written back what the last
                        // slot produced up to 4 steps before that, and its Nyquist bins (program order + the wait below)
                        const int done_rows = t0 - NSLOTS * LAG - 8;       // (the write-back below trails the last slot by one pair)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores of the previous block
                        if (done_rows > 0 && lane == 0)
                            __hip_atomic_store(my_progress, (unsigned)done_rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        wait_rows(t0 + 16 + 40);   // the loader fetches up to row t0 + 16 in this block, the Nyquist loader a frame ahead
                    }
                }
                if (r_service) {
                    LWS_SETPRIO(3);   // (and back to 0 with everybody else after the publish below)
                    // Nyquist bins of the frames that ended at phase 0 of this block (every slot has published bin C-1 now)
#if LWS_Q8
                    if (PA == RE && hf == 0)
                        service_nyquist_rows<Q, L, MULTI, H16, RE>(a, sv, lane, t0, wg, n_eff, n_groups, thr_eff, state_nyq_b, amp_nyq_b);
#else
                    constexpr bool nyq_split = (RE & 2) != 0;   // frames ending in the second pair of a quad: see service_nyquist
                    if (PA == RE && hf == 0)   // (RE != 0: the frames end at phase RE of the block)
                        service_nyquist<Q, L, MASK, MULTI, H16, RE, (nyq_split ? 2 : 0)>(a, sv, lane, t0, wg, n_eff, n_groups, thr_eff, state_nyq_b, amp_nyq_b);
                    if (nyq_split && PA == RE - 2 && hf == 0)
                        service_nyquist<Q, L, MASK, MULTI, H16, RE, 1>(a, sv, lane, t0, wg, n_eff, n_groups, thr_eff, state_nyq_b, amp_nyq_b);
#endif
                    // loader: feed set 0 with the values the virtual previous sweep would produce at clocks PA, PA+1
                    // (the loader is sweep slot -1: its lanes sit at bin (t0 - 8*lane) mod 512 of their frames)
                    int ldb[NBLK], ldu[NBLK], ldh[NBLK];
    #pragma unroll
                    for (int m = 0; m < NBLK; ++m) {
                        ldu[m] = blk_mod((t0 >> 3) - m) * BLK_BYTES;
                        ldb[m] = ldu[m] + rl * LANE_B;
                        ldh[m] = ldb[m] + cx.halo_shift;
                    }
                    const int cb0 = (t0 - SKEW * rl) & (ROWP - 1);
                    const bool l_st = cb0 == 0, l_en = cb0 == C - th0(RE), l_en1 = (RE != 0) && cb0 == C - th1(RE);
                    const float2 vA = raw_value<H16>(make_float2(amp_cur[PA], amp_nxt[PA]));
                    const float2 vB = raw_value<H16>(make_float2(amp_cur[PA + 1], amp_nxt[PA + 1]));
                    lds_write128(ring_addr<PA, 0>(ldb), vA, vB);   // one ring cell
                    lds_write128(ring_addr<PA, 0>(ldh), vA, vB);   // (halo copy, or the same cell again)
                    image_publish<L, PA, PA, 0, RE>(ldu, l_st, l_en, cx.dummy, vA, l_en1);
                    image_publish<L, PA + 1, PA + 1, 0, RE>(ldu, l_st, l_en, cx.dummy, vB, l_en1);
                    // write-back: the two values the last sweep slot produced in the previous pair, steps t0+PA-2 and t0+PA-1 (one
                    // ring cell of its output set; complete, every slot has finished that pair) go to the rows of its clock.  Done
                    // here, by the wave with time to spare, so that the sweep slots carry no store and no branch around one.
                    {
                        if constexpr (PA == 0) {   // where the last slot's lanes are in this block of its clock
                            wb_prev = wb_cur;
                            const int vv = t0 - NSLOTS * LAG - SKEW * rl;
                            const int kap = vv >> ROWP_SHIFT;
                            const int gl = (int)(((float)kap + 0.5f) * inv_kr), k = kap - gl * Kr;
                            wb_cur = (vv >= 0) && ((vv & (ROWP - 1)) < C) && (k * ROWL + rl < a.Tp) && (gl * nwg + wg < n_groups);
                        }
                        const v4f w = lds_read128(ring_addr<PA, -2>(ldb) + NSLOTS * SET_BYTES);
                        int r0 = SPW > 1 ? wmod + PA - 2 : tmod + PA - 2 - NSLOTS * LAG;
                        r0 += (r0 < 0) ? G : 0;
                        if constexpr (SPW == 1) r0 += (r0 < 0) ? G : 0;        // (G >= 512 > NSLOTS * LAG / 2)
                        int r1 = r0 + 1;
                        r1 -= (r1 >= G) ? G : 0;
                        if (PA == 0 ? wb_prev : wb_cur) {
                            store_l2<H16>(state_w_b, (size_t)r0 * ROWL + rl, make_float2(w.x, w.y), MULTI);
                            store_l2<H16>(state_w_b, (size_t)r1 * ROWL + rl, make_float2(w.z, w.w), MULTI);
                        }
                    }
                    int i0 = tmod + PA + 8, i1 = tmod + PA + 9;
                    i0 -= (i0 >= G) ? G : 0;
                    i1 -= (i1 >= G) ? G : 0;
                    const float2 p0 = load_l2<H16>(state_w_b, (size_t)i0 * ROWL + rl);
                    const float2 p1 = load_l2<H16>(state_w_b, (size_t)i1 * ROWL + rl);
                    amp_cur[PA] = p0.x; amp_nxt[PA] = p0.y;
                    amp_cur[PA + 1] = p1.x; amp_nxt[PA + 1] = p1.y;
                }
                flow_publish(lane, wave, t0 + PA + 2);
                LWS_SETPRIO(0);   // polling for the next pair must not take issue slots from the wave still working
            });
            if (r_compute)   // the next block's magnitudes (issued 4 pairs ago) are in their registers before anything may move them
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(amp_nxt[0]), "+v"(amp_nxt[1]), "+v"(amp_nxt[2]), "+v"(amp_nxt[3]),
                             "+v"(amp_nxt[4]), "+v"(amp_nxt[5]), "+v"(amp_nxt[6]), "+v"(amp_nxt[7]) : : "memory");
        }
    };
#if LWS_Q8
    if (is_compute) role_loop(std::integral_constant<int, 0>{});
    else if (is_helper && helper_no == 1) role_loop(std::integral_constant<int, 1>{});
    else if (is_helper) role_loop(std::integral_constant<int, NHELP>{});
    else role_loop(std::integral_constant<int, 9>{});
#elif LWS_SPLIT_ROLES
    if (is_compute) role_loop(std::integral_constant<int, 0>{});
    else role_loop(std::integral_constant<int, 9>{});
#else
    role_loop(std::integral_constant<int, -1>{});
#endif
}

// ---------------------------------------------------------------------------------------------
// layout conversion: reference extended layout <-> time-skewed layout
// ---------------------------------------------------------------------------------------------
// Both directions move 64 x 64 tiles (64 frames of one lane round x 64 production times) through LDS: in the reference
// layout a frame's bins are contiguous, in the skewed layout the 64 lanes of a time step are, so the tile is read along
// one and written along the other and both sides of the copy are full 512-byte segments.
// grid: (Kt * NT, B) with Kt = ceil(Tp / 64) groups of 64 frames and NT = ceil((SKEW*63 + C) / 64) time tiles per group;
// 256 threads.  Frame me sits in column me mod ROWL of the rows (SKEW*me + c) mod G: with two waves per sweep slot
// (wide build) a ring row, hence a row of the layout, is 128 frames wide and two groups of 64 frames share its rows.
//
// fp16 storage (H16): the values are stored multiplied by store_scale(largest magnitude), so that magnitude has to be
// known before the first store -- a reduction pass of its own (k_amax_*) instead of the atomic maximum the fp32 kernels
// take on the way.  On the way out the state only contributes its PHASE: a bin that some sweep could have updated is
// returned with the fp32 target magnitude it was given (the caller's |S|), a bin no sweep could touch keeps the caller's
// value bit for bit; so fp16 storage costs phase accuracy, never magnitude accuracy.
constexpr int TILE = 64, TPAD = TILE + 1;

// |S| of a complex64 input exactly as k_prep forms it (fp64 square root, rounded once)
__device__ __forceinline__ float mag_of(float2 v, double *mag64 = nullptr) {
    const double m = sqrt((double)v.x * (double)v.x + (double)v.y * (double)v.y);
    if (mag64) *mag64 = m;
    return (float)m;
}

// largest magnitude of the real frames of each spectrogram; grid (blocks, B)
__global__ void __launch_bounds__(256) k_amax_in(const float2 *in, unsigned *amax_bits, size_t n_per) {
    __shared__ float red[256];
    const float2 *p = in + (size_t)blockIdx.y * n_per;
    float mx = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_per; i += (size_t)gridDim.x * 256) mx = fmaxf(mx, mag_of(p[i]));
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s2]);
        __syncthreads();
    }
    if (threadIdx.x == 0 && red[0] > 0.f) atomicMax(amax_bits + blockIdx.y, __float_as_uint(red[0]));
}
// the same from the extended magnitude buffer [B][Tp][Np] (real frames only; the pad columns repeat real bins)
__global__ void __launch_bounds__(256) k_amax_ext(const float *amp, unsigned *amax_bits, int T, int Np, int Q) {
    __shared__ float red[256];
    const int Tp = T + 2 * (Q - 1);
    const float *p = amp + ((size_t)blockIdx.y * Tp + (Q - 1)) * Np;
    const size_t n_per = (size_t)T * Np;
    float mx = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_per; i += (size_t)gridDim.x * 256) mx = fmaxf(mx, p[i]);
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s2]);
        __syncthreads();
    }
    if (threadIdx.x == 0 && red[0] > 0.f) atomicMax(amax_bits + blockIdx.y, __float_as_uint(red[0]));
}
// smallest threshold of each spectrogram that any of its bins exceeds (+inf if none): a bin whose stored magnitude is not
// above it is never updated by this call
__global__ void __launch_bounds__(64) k_thr_min(const float *thr, const float *amax, int n_iters, float *thr_min) {
    const int b = blockIdx.x;
    float m = __builtin_inff();
    for (int i = threadIdx.x; i < n_iters; i += 64) {
        const float th = thr[(size_t)b * n_iters + i];
        if (amax[b] > th) m = fminf(m, th);
    }
    for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o));
    if (threadIdx.x == 0) thr_min[b] = m;
}
// fp16 storage: could any sweep of the call have updated a bin of fp32 magnitude `a`?  (the kernel's own test, on the stored
// half value in the scaled domain)
__device__ __forceinline__ bool ever_active_h16(float a, float sc, float thr_min) { return unpack_h(pack_h(a * sc)) > thr_min * sc; }
// fp16 storage: phase of the stored value v, magnitude a
__device__ __forceinline__ float2 with_magnitude(float2 v, float a, bool *ok) {
    const float m2 = v.x * v.x + v.y * v.y;
    *ok = m2 > 0.f;
    const float s = a / sqrtf(m2);
    return make_float2(v.x * s, v.y * s);
}

template <bool H16>
__global__ void __launch_bounds__(256) k_to_skew(const float2 *state, const float *amp, void *state_w_, void *amp_w_,
                                                  void *state_nyq_, void *amp_nyq_, unsigned *amax_bits, int T, int F,
                                                  int L, int Q, int G, int TpPad, int NT, const int *gate) {
    using ST = Store<H16>;
    __shared__ float2 ts[TILE][TPAD];
    __shared__ float ta[TILE][TPAD];
    __shared__ float red[256];
    if (gate != nullptr && *gate == 0) return;
    const int kk = blockIdx.x / NT, tt = blockIdx.x - kk * NT, b = blockIdx.y;
    const int Np = F + 2 * L, Tp = T + 2 * (Q - 1), C = F - 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tau0 = SKEW * LANES * kk + TILE * tt;          // first production time of the tile (before the mod G)
    typename ST::cplx *sw = static_cast<typename ST::cplx *>(state_w_) + (size_t)b * G * ROWL;
    typename ST::real *aw = static_cast<typename ST::real *>(amp_w_) + (size_t)b * G * ROWL;
    typename ST::cplx *snq = static_cast<typename ST::cplx *>(state_nyq_) + (size_t)b * TpPad;
    typename ST::real *anq = static_cast<typename ST::real *>(amp_nyq_) + (size_t)b * TpPad;
    const float sc = H16 ? store_scale(__uint_as_float(amax_bits[b])) : 1.f;   // (H16: k_amax_ext ran before)
    float mx = 0.f;
    float2 vin[TILE / 4];
    float ain[TILE / 4];
#pragma unroll
    for (int i = 0; i < TILE / 4; ++i) {                     // one frame per wave: 64 consecutive bins; all loads first
        const int ml = wave + 4 * i;
        const int me = LANES * kk + ml, c = tau0 + lane - SKEW * me;   // tile column = production time - tau0
        const bool in_range = me < Tp && c >= 0 && c < C;
        const size_t idx = ((size_t)b * Tp + me) * Np + L + c;
        vin[i] = in_range ? state[idx] : make_float2(0.f, 0.f);
        ain[i] = in_range ? amp[idx] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TILE / 4; ++i) {
        const int ml = wave + 4 * i;
        const int me = LANES * kk + ml;
        if (me >= Q - 1 && me < T + Q - 1) mx = fmaxf(mx, ain[i]);
        ts[ml][lane] = vin[i];
        ta[ml][lane] = ain[i];
    }
    if (tt == 0 && wave == 0) {                              // Nyquist bins of the round's frames
        const int me = LANES * kk + lane;
        if (me < Tp) {
            const size_t i = ((size_t)b * Tp + me) * Np + L + C;
            const float av = amp[i];
            const float2 v = state[i];
            if constexpr (H16) { snq[me] = pack_h2(make_float2(v.x * sc, v.y * sc)); anq[me] = pack_h(av * sc); }
            else { snq[me] = v; anq[me] = av; }
            if (me >= Q - 1 && me < T + Q - 1) mx = fmaxf(mx, av);
        }
    }
    __syncthreads();
    for (int tl = wave; tl < TILE; tl += 4) {                // one production time per wave: 64 consecutive lanes
        const int me = LANES * kk + lane, c = tau0 + tl - SKEW * me;
        if (me < Tp && c >= 0 && c < C) {
            const size_t idx = (size_t)((tau0 + tl) % G) * ROWL + (me & (ROWL - 1));
            const float2 v = ts[lane][tl];
            if constexpr (H16) { sw[idx] = pack_h2(make_float2(v.x * sc, v.y * sc)); aw[idx] = pack_h(ta[lane][tl] * sc); }
            else { sw[idx] = v; aw[idx] = ta[lane][tl]; }
        }
    }
    if constexpr (!H16) {
        // block max -> atomic max on the bit pattern (non-negative floats order like unsigned ints)
        red[threadIdx.x] = mx;
        __syncthreads();
        for (int s2 = 128; s2 > 0; s2 >>= 1) {
            if (threadIdx.x < s2) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s2]);
            __syncthreads();
        }
        if (threadIdx.x == 0 && red[0] > 0.f) atomicMax(amax_bits + b, __float_as_uint(red[0]));
    }
}

// Also restores the Hermitian pad columns and the Nyquist column of the extended layout.
template <bool H16>
__global__ void __launch_bounds__(256) k_from_skew(float2 *state, const float *amp, const void *state_w_, const void *state_nyq_,
                                                    const float *amax, const float *thr_min, int T, int F, int L, int Q, int G,
                                                    int TpPad, int NT) {
    using ST = Store<H16>;
    __shared__ float2 ts[TILE][TPAD];
    const int kk = blockIdx.x / NT, tt = blockIdx.x - kk * NT, b = blockIdx.y;
    const int Np = F + 2 * L, Tp = T + 2 * (Q - 1), C = F - 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tau0 = SKEW * LANES * kk + TILE * tt;
    const typename ST::cplx *sw = static_cast<const typename ST::cplx *>(state_w_) + (size_t)b * G * ROWL;
    const typename ST::cplx *snq = static_cast<const typename ST::cplx *>(state_nyq_) + (size_t)b * TpPad;
    const float sc = H16 ? store_scale(amax[b]) : 1.f, tmin = H16 ? thr_min[b] : 0.f;
    float2 vin[TILE / 4];
#pragma unroll
    for (int i = 0; i < TILE / 4; ++i) {           // all loads first
        const int tl = wave + 4 * i;
        const int me = LANES * kk + lane, c = tau0 + tl - SKEW * me;
        const bool in_range = me < Tp && c >= 0 && c < C;
        if constexpr (H16) vin[i] = in_range ? unpack_h2(sw[(size_t)((tau0 + tl) % G) * ROWL + (me & (ROWL - 1))]) : make_float2(0.f, 0.f);
        else vin[i] = in_range ? sw[(size_t)((tau0 + tl) % G) * ROWL + (me & (ROWL - 1))] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < TILE / 4; ++i) ts[lane][wave + 4 * i] = vin[i];
    __syncthreads();
    for (int ml = wave; ml < TILE; ml += 4) {
        const int me = LANES * kk + ml, c = tau0 + lane - SKEW * me;
        if (me < Tp && c >= 0 && c < C) {
            float2 *orow = state + ((size_t)b * Tp + me) * Np;
            float2 v = ts[ml][lane];
            if constexpr (H16) {     // phase from the fp16 state, magnitude from the fp32 target; untouched bins stay as they are
                const float a = amp[((size_t)b * Tp + me) * Np + L + c];
                bool ok;
                v = with_magnitude(v, a, &ok);
                if (!(ok && me >= Q - 1 && me < T + Q - 1 && ever_active_h16(a, sc, tmin))) continue;
            }
            orow[L + c] = v;
            const float2 vc = make_float2(v.x, -v.y);
            if (c >= 1 && c <= L) orow[L - c] = vc;                 // image below DC
            if (c >= C - L) orow[L + 2 * C - c] = vc;               // image above Nyquist
        }
    }
    if (tt == 0 && wave == 0) {
        const int me = LANES * kk + lane;
        if (me < Tp) {
            const size_t i = ((size_t)b * Tp + me) * Np + L + C;
            if constexpr (H16) {
                const float a = amp[i];
                bool ok;
                const float2 v = with_magnitude(unpack_h2(snq[me]), a, &ok);
                if (ok && me >= Q - 1 && me < T + Q - 1 && ever_active_h16(a, sc, tmin)) state[i] = v;
            } else {
                state[i] = snq[me];
            }
        }
    }
}

// The same two conversions straight from / to the caller's unpadded [B][T][F] complex64 spectrograms, for calls that
// are one batch stage: what k_prep + k_to_skew and k_from_skew + k_extract do in two passes each.
// partial: [B][tiles] sums of |S| over the real frames in fp64, one per tile (block), for mean|S|.
template <bool H16>
__global__ void __launch_bounds__(256) k_in_to_skew(const float2 *in, void *state_w_, void *amp_w_, void *state_nyq_,
                                                     void *amp_nyq_, unsigned *amax_bits, double *partial, int T, int F,
                                                     int Q, int G, int TpPad, int NT, const int *gate) {
    using ST = Store<H16>;
    __shared__ float2 ts[TILE][TPAD];
    __shared__ float ta[TILE][TPAD];
    __shared__ float red[256];
    __shared__ double dred[256];
    if (gate != nullptr && *gate == 0) return;
    const int kk = blockIdx.x / NT, tt = blockIdx.x - kk * NT, b = blockIdx.y;
    const int Tp = T + 2 * (Q - 1), C = F - 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tau0 = SKEW * LANES * kk + TILE * tt;
    typename ST::cplx *sw = static_cast<typename ST::cplx *>(state_w_) + (size_t)b * G * ROWL;
    typename ST::real *aw = static_cast<typename ST::real *>(amp_w_) + (size_t)b * G * ROWL;
    typename ST::cplx *snq = static_cast<typename ST::cplx *>(state_nyq_) + (size_t)b * TpPad;
    typename ST::real *anq = static_cast<typename ST::real *>(amp_nyq_) + (size_t)b * TpPad;
    const float sc = H16 ? store_scale(__uint_as_float(amax_bits[b])) : 1.f;   // (H16: k_amax_in ran before)
    float mx = 0.f;
    double msum = 0.0;                                       // this thread's share of sum |S| over the real frames
    // all of a thread's loads are issued before the first one is used: 16 x 512 bytes in flight per wave
    float2 vin[TILE / 4];
#pragma unroll
    for (int i = 0; i < TILE / 4; ++i) {
        const int ml = wave + 4 * i;
        const int me = LANES * kk + ml, c = tau0 + lane - SKEW * me;
        int src = me - (Q - 1);                                  // edge-pad frames repeat the first / last frame
        src = src < 0 ? 0 : (src > T - 1 ? T - 1 : src);
        vin[i] = (me < Tp && c >= 0 && c < C) ? in[((size_t)b * T + src) * F + c] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < TILE / 4; ++i) {
        const int ml = wave + 4 * i;
        const int me = LANES * kk + ml, c = tau0 + lane - SKEW * me;
        const bool real_frame = me >= Q - 1 && me < T + Q - 1;
        const float2 v = vin[i];
        float av = 0.f;
        if (me < Tp && c >= 0 && c < C) {
            double mag;
            av = mag_of(v, &mag);                                // as k_prep: |S| in fp64, then rounded
            if (real_frame) { mx = fmaxf(mx, av); msum += mag; }
        }
        ts[ml][lane] = v;
        ta[ml][lane] = av;
    }
    if (tt == 0 && wave == 0) {                                  // Nyquist bins of the round's frames
        const int me = LANES * kk + lane;
        if (me < Tp) {
            int src = me - (Q - 1);
            src = src < 0 ? 0 : (src > T - 1 ? T - 1 : src);
            const float2 v = in[((size_t)b * T + src) * F + C];
            double mag;
            const float av = mag_of(v, &mag);
            if constexpr (H16) { snq[me] = pack_h2(make_float2(v.x * sc, v.y * sc)); anq[me] = pack_h(av * sc); }
            else { snq[me] = v; anq[me] = av; }
            if (me >= Q - 1 && me < T + Q - 1) {
                mx = fmaxf(mx, av);
                msum += mag;
            }
        }
    }
    __syncthreads();
    for (int tl = wave; tl < TILE; tl += 4) {
        const int me = LANES * kk + lane, c = tau0 + tl - SKEW * me;
        if (me < Tp && c >= 0 && c < C) {
            const size_t idx = (size_t)((tau0 + tl) % G) * ROWL + (me & (ROWL - 1));
            const float2 v = ts[lane][tl];
            if constexpr (H16) { sw[idx] = pack_h2(make_float2(v.x * sc, v.y * sc)); aw[idx] = pack_h(ta[lane][tl] * sc); }
            else { sw[idx] = v; aw[idx] = ta[lane][tl]; }
        }
    }
    red[threadIdx.x] = mx;
    dred[threadIdx.x] = msum;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {               // fixed tree: deterministic
        if (threadIdx.x < s2) {
            red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s2]);
            dred[threadIdx.x] += dred[threadIdx.x + s2];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (!H16 && red[0] > 0.f) atomicMax(amax_bits + b, __float_as_uint(red[0]));
        partial[(size_t)b * gridDim.x + blockIdx.x] = dred[0];   // one partial sum per tile
    }
}

// mean|S| of each spectrogram from the partial sums, fixed order
__global__ void __launch_bounds__(256) k_mean_partials(const double *partial, double *mean_amp, int n, double denom) {
    __shared__ double red[256];
    const int b = blockIdx.x;
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += partial[(size_t)b * n + i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0) mean_amp[b] = red[0] / denom;
}

// skewed layout -> unpadded [B][T][F] output (real frames only, no pad columns).  `in`: the caller's input (fp16 storage
// only; may be the same buffer as `out`: every element is read and written by the same thread).
template <bool H16>
__global__ void __launch_bounds__(256) k_skew_to_out(float2 *out, const float2 *in, const void *state_w_, const void *state_nyq_,
                                                      const float *amax, const float *thr_min, int T, int F, int Q, int G,
                                                      int TpPad, int NT) {
    using ST = Store<H16>;
    __shared__ float2 ts[TILE][TPAD];
    const int kk = blockIdx.x / NT, tt = blockIdx.x - kk * NT, b = blockIdx.y;
    const int Tp = T + 2 * (Q - 1), C = F - 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tau0 = SKEW * LANES * kk + TILE * tt;
    const typename ST::cplx *sw = static_cast<const typename ST::cplx *>(state_w_) + (size_t)b * G * ROWL;
    const typename ST::cplx *snq = static_cast<const typename ST::cplx *>(state_nyq_) + (size_t)b * TpPad;
    const float sc = H16 ? store_scale(amax[b]) : 1.f, tmin = H16 ? thr_min[b] : 0.f;
    float2 vin[TILE / 4];
#pragma unroll
    for (int i = 0; i < TILE / 4; ++i) {           // all loads first
        const int tl = wave + 4 * i;
        const int me = LANES * kk + lane, c = tau0 + tl - SKEW * me;
        const bool in_range = me < Tp && c >= 0 && c < C;
        if constexpr (H16) vin[i] = in_range ? unpack_h2(sw[(size_t)((tau0 + tl) % G) * ROWL + (me & (ROWL - 1))]) : make_float2(0.f, 0.f);
        else vin[i] = in_range ? sw[(size_t)((tau0 + tl) % G) * ROWL + (me & (ROWL - 1))] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < TILE / 4; ++i) ts[lane][wave + 4 * i] = vin[i];
    __syncthreads();
    for (int ml = wave; ml < TILE; ml += 4) {
        const int me = LANES * kk + ml, c = tau0 + lane - SKEW * me;
        if (me >= Q - 1 && me < T + Q - 1 && c >= 0 && c < C) {
            const size_t o = ((size_t)b * T + (me - (Q - 1))) * F + c;
            if constexpr (H16) {
                const float2 orig = in[o];
                const float a = mag_of(orig);
                bool ok;
                const float2 v = with_magnitude(ts[ml][lane], a, &ok);
                out[o] = (ok && ever_active_h16(a, sc, tmin)) ? v : orig;
            } else {
                out[o] = ts[ml][lane];
            }
        }
    }
    if (tt == 0 && wave == 0) {
        const int me = LANES * kk + lane;
        if (me >= Q - 1 && me < T + Q - 1) {
            const size_t o = ((size_t)b * T + (me - (Q - 1))) * F + C;
            if constexpr (H16) {
                const float2 orig = in[o];
                const float a = mag_of(orig);
                bool ok;
                const float2 v = with_magnitude(unpack_h2(snq[me]), a, &ok);
                out[o] = (ok && ever_active_h16(a, sc, tmin)) ? v : orig;
            } else {
                out[o] = snq[me];
            }
        }
    }
}

constexpr uint64_t mask_all(int Q, int L) { return (1ull << (Q * (L + 1))) - 1ull; }

template <int Q, int L, uint64_t MASK, bool MULTI, bool H16, int RE> hipError_t launch_km(const SysArgs &a, int grid, hipStream_t s) {
    static std::atomic<unsigned long long> attr_set{0};   // one bit per device
    int attr_dev;
    if (lws::attr_needed(attr_set, &attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_systolic<Q, L, MASK, MULTI, H16, RE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        lws::attr_done(attr_set, attr_dev);
    }
    if constexpr (MULTI) {
        // The workgroups of a spectrogram wait for each other: all of the grid must be resident at once.  prepare() sizes the
        // grid to at most one workgroup per CU; here the other half of that argument is checked against the runtime's own
        // occupancy figure for this very kernel (once per kernel and process): a grid it cannot hold is refused, not launched.
        // (What no query sees -- a device shared with another process -- is covered by the hand-over time-out and the
        // device-side re-run with one workgroup per spectrogram, run_kernel.)
        static std::atomic<int> per_cu[64];             // per device (zero-initialised: 0 = not asked yet, else figure + 1)
        int dev = 0, n_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return hipErrorUnknown;
        std::atomic<int> &slot = per_cu[dev & 63];
        int occ = slot.load(std::memory_order_relaxed) - 1;
        if (occ < 0) {
            int q = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, k_systolic<Q, L, MASK, MULTI, H16, RE>, NTHREADS, LDS_BYTES) != hipSuccess) q = 0;
            slot.store(q + 1, std::memory_order_relaxed);
            occ = q;
        }
        if (occ < 1 || (long)grid > (long)occ * n_cu) return hipErrorLaunchOutOfResources;
    }
    hipLaunchKernelGGL((k_systolic<Q, L, MASK, MULTI, H16, RE>), dim3(grid), dim3(NTHREADS), LDS_BYTES, s, a);
    return hipGetLastError();
}
template <int Q, int L, uint64_t MASK, int RE> hipError_t launch_kr(const SysArgs &a, int grid, bool h16, hipStream_t s) {
#if LWS_TW && LWS_Q8
    if (h16) return hipErrorInvalidValue;   // (not instantiated: systolic_build refuses fp16 storage for this build)
#else
    if (h16) return a.nwg > 1 ? launch_km<Q, L, MASK, true, true, RE>(a, grid, s) : launch_km<Q, L, MASK, false, true, RE>(a, grid, s);
#endif
    return a.nwg > 1 ? launch_km<Q, L, MASK, true, false, RE>(a, grid, s) : launch_km<Q, L, MASK, false, false, RE>(a, grid, s);
}
// one build of the kernel per phase of the block at which the frames end, (F-1) mod 8 (see th0)
template <int Q, int L, uint64_t MASK> hipError_t launch_k(const SysArgs &a, int grid, bool h16, hipStream_t s) {
#if LWS_Q8 && !LWS_TW
    // (the hop is an eighth of the frame size: F-1 is a multiple of 4, the frames end at phase 0 or 4)
    if ((a.C & 7) == 4) return launch_kr<Q, L, MASK, 4>(a, grid, h16, s);
    if ((a.C & 7) != 0) return hipErrorInvalidValue;
#else
    switch (a.C & 7) {
    case 2: return launch_kr<Q, L, MASK, 2>(a, grid, h16, s);
    case 4: return launch_kr<Q, L, MASK, 4>(a, grid, h16, s);
    case 6: return launch_kr<Q, L, MASK, 6>(a, grid, h16, s);
    default: break;
    }
#endif
    return launch_kr<Q, L, MASK, 0>(a, grid, h16, s);
}

#if !LWS_Q8 && !LWS_L7 && !LWS_TW
// mask bit r*(L+1)+k set <=> |W[0][r][k]| > 1e-12.  Default sqrt-Hann windows give these patterns (L = 5):
constexpr uint64_t MASK_Q4_L5_DEFAULT = 0b111111'010111'111111'000011u;  // (r=3 | r=2 | r=1 | r=0), 6 bits each, bit k: r=0:{0,1} r=1:all r=2:{0,1,2,4} r=3:all
constexpr uint64_t MASK_Q2_L5_DEFAULT = 0b010111'000011u;                            // r=0:{0,1} r=1:{0,1,2,4}
#endif

struct Tables {
    int Q, L;
    uint64_t mask;
    bool k0real, r13;  // structure the kernels can exploit (FLAG_K0REAL / FLAG_R13)
    // W[p][r][k] = W[0][r][k] exp(2 pi j p r tw_s / tw_P); LWS_TW: the table of the kernel on the device
    // (these fields sit at the same offsets in every build of this file -- w[] below does not have the same length in all)
    int tw_P = 0, tw_s = 0;
    float *tw_dev = nullptr;
    float tw_nyq[16] = {0};  // tau_r(F-1), r = 0..7
    float w[2 * NW];   // (re, im) in the order of SysArgs::w
};

}  // namespace

// =============================================================================================
// host side
// =============================================================================================
hipError_t systolic_build(SystolicPlan &sp, int F, int Lu, int Q, int Qp, const double *const W[3], bool fp16_storage) {
    // Lu: the caller's stencil half-width (its weight tensors have Lu + 1 columns, its extended buffers 2 Lu pad columns);
    // L: that of the kernel build -- the next odd number, the extra tap with weight zero (its mask bit is clear: never fetched)
    const int L = (LWS_Q8 || LWS_TW) ? (Lu <= 5 ? 5 : Lu) : (Lu | 1), K1u = Lu + 1;   // (the Q = 8 and table-twiddle builds exist for L = 5 only: narrower stencils run on them)
    sp.F = F; sp.L = Lu; sp.Lk = L; sp.Q = Q; sp.h16 = fp16_storage;
    for (int i = 0; i < 3; ++i) sp.ok[i] = false;
    if (Lu < 0) return hipSuccess;
    const int C = F - 1;
    // stencil half-widths: L = 5 (every default configuration) with the specialised tap masks, L = 3 and 1 with all taps.  The
    // kernel's L is odd (the tap windows are fetched as aligned pairs of bins) and at most SKEW - 3: a lane works on two bins per
    // rendez-vous, so the newest tap of the pair's second bin, (m-1, c+1+L), must be at least two steps old when the pair
    // starts (L = 7 is not: generic engine).
    // Qp: rows of the weight tensors -- Q (summarised: row = bin mod Q) or N = 2 (F-1) (general: row = bin; use_simplifications =
    // False or a hop that does not divide the frame, lws.pyx:164-168).  Either way the kernels want the twiddle structure, below.
    if (Qp != Q && Qp != 2 * (F - 1)) return hipSuccess;
#if LWS_TW && LWS_Q8
    if (Q < 5 || Q > 8 || Lu > 5 || fp16_storage) return hipSuccess;   // (fp32 storage only: half the instantiations of the other builds)
    if (LWS_TWQ && Q != LWS_TWQ) return hipSuccess;                     // (the shallower rings: exactly that many frames per stencil row)
#elif LWS_TW
    if (Q < 2 || Q > 4 || Lu > 5) return hipSuccess;     // (stencils narrower than L = 5 run with zero weights; Q = 2: a hop above half the frame)
#elif LWS_Q8
    if (Q != 8 || L != 5) return hipSuccess;
#elif LWS_R16
    if (Q != 2 || !(L == 5 || L == 3 || L == 1)) return hipSuccess;
#elif LWS_L7
    if (!(Q == 2 || Q == 4) || L != 7) return hipSuccess;      // (Lu = 6: the extra tap has weight zero)
#else
    if (!(Q == 2 || Q == 4) || !(L == 5 || L == 3 || L == 1)) return hipSuccess;
#endif
    // F-1 even (a pair of bins never straddles bin C); not a multiple of 8: the frames end inside a block (th0), and the block
    // before that one must not be the frame's first
    if ((C & 1) != 0 || C > ROWP || C < 16 || (C % 8 != 0 && (C < 24 || (LWS_Q8 && !LWS_TW && C % 8 != 4))) || (LWS_L7 && C > 512)) return hipSuccess;
    if ((Q - 1) * SKEW + L + 1 > LAG) return hipSuccess;
    const int K1 = L + 1;
    for (int i = 0; i < 3; ++i) {
        if (!W[i]) continue;
        // twiddle structure: W[p][r][k] == W[0][r][k] * exp(2j*pi*p*r*s/P) for every row p of the tensor
        double scale = 0;
        for (size_t x = 0; x < (size_t)Q * Q * K1u; ++x) scale = std::fmax(scale, std::hypot(W[i][2 * x], W[i][2 * x + 1]));
        // W[p][r][k] of the caller's tensor; zero for the tap an even Lu does not have
        auto wre = [&](int p, int r, int k) { return k <= Lu ? W[i][2 * ((p * Q + r) * K1u + k)] : 0.0; };
        auto wim = [&](int p, int r, int k) { return k <= Lu ? W[i][2 * ((p * Q + r) * K1u + k) + 1] : 0.0; };
        int twP = 0, twS = 0;
        if (!lws::weights_twiddle(W[i], Q, Qp, Lu, TW ? TW_PMAX : Q, &twP, &twS)) continue;   // (lws_online.hip: every row p of the tensor, in fp64)
        if (twP == 0) { twP = Q; twS = 1; }   // (no neighbour-frame weights at all: any twiddle will do)
#if LWS_TW
        // (tensors whose twiddles are the eighth turns of the static builds are theirs: tried before this one, lws_capi.hip)
#else
        // the static builds: the twiddle of a bin is exp(2 pi j (bin mod Q) r / Q), compiled into the unrolled step loop
        if (twP != Q || twS != 1) continue;
#endif
        Tables *tb = new Tables();
        tb->Q = Q; tb->L = L; tb->mask = 0;
        tb->tw_P = twP; tb->tw_s = twS;
#if LWS_TW
        {
            // the kernel's table: row p (p = 0 .. P + 7, periodic) = tau_1(p), tau_2(p), tau_3(p), 0; formed in fp64, rounded once
            constexpr int RF = TW_ROW / 4;     // floats per table row
            std::vector<float> tab((size_t)(twP + 8) * RF, 0.f);
            for (int pp = 0; pp < twP + 8; ++pp)
                for (int r = 1; r < Q; ++r) {
                    const double ang = 2.0 * M_PI * (double)(((long long)pp * r * twS) % twP) / twP;
                    tab[(size_t)pp * RF + 2 * (r - 1)] = (float)std::cos(ang);
                    tab[(size_t)pp * RF + 2 * (r - 1) + 1] = (float)std::sin(ang);
                }
            for (int r = 0; r < 8; ++r) {
                const double ang = 2.0 * M_PI * (double)(((long long)C * r * twS) % twP) / twP;
                tb->tw_nyq[2 * r] = (float)std::cos(ang);
                tb->tw_nyq[2 * r + 1] = (float)std::sin(ang);
            }
            hipError_t e = hipMalloc(&tb->tw_dev, tab.size() * sizeof(float));
            if (e == hipSuccess) e = hipMemcpy(tb->tw_dev, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice);
            if (e != hipSuccess) { if (tb->tw_dev) (void)hipFree(tb->tw_dev); delete tb; return e; }
        }
#endif
        for (int r = 0; r < Q; ++r)
            for (int k = 0; k <= L; ++k) {
                const double wr = wre(0, r, k), wi = wim(0, r, k);
                const bool on = std::hypot(wr, wi) > 1.0e-12;  // lws.pyx:231-232
                if (on) tb->mask |= 1ull << (r * K1 + k);
#if LWS_Q8
                // per-wave lists (widx: position in the list of the wave that sums frames m-+r) and the full set 0 for the
                // Nyquist lanes; second set: W[0][r][k] exp(j pi / 4), formed in fp64 and rounded once
                const double h = std::sqrt(0.5);
                const int x0 = row_owner(r) * WLIST + widx<8, 5>(0, r, k), xn = WNYQ + r * K1 + k;
                tb->w[2 * x0] = tb->w[2 * xn] = on ? (float)wr : 0.f;
                tb->w[2 * x0 + 1] = tb->w[2 * xn + 1] = on ? (float)wi : 0.f;
                if (r & 1) {
                    const int x1 = row_owner(r) * WLIST + widx<8, 5>(1, r, k);
                    tb->w[2 * x1] = on ? (float)((wr - wi) * h) : 0.f;
                    tb->w[2 * x1 + 1] = on ? (float)((wr + wi) * h) : 0.f;
                }
#else
                tb->w[2 * (r * K1 + k)] = on ? (float)wr : 0.f;
                tb->w[2 * (r * K1 + k) + 1] = on ? (float)wi : 0.f;
#endif
            }
        // structure of symmetric windows, checked on the fp64 weights well below fp32 resolution
        tb->k0real = true;
        for (int r = 1; r < Q; ++r)
            if (((tb->mask >> (r * K1)) & 1ull) && std::fabs(wim(0, r, 0)) > 1e-13 * scale) tb->k0real = false;
        tb->r13 = (Q == 4);
        for (int k = 2; k <= L && tb->r13; ++k) {
            const bool on1 = (tb->mask >> (1 * K1 + k)) & 1u, on3 = (tb->mask >> (3 * K1 + k)) & 1u;
            if (on1 != on3) { tb->r13 = false; break; }
            if (!on1) continue;
            double xr = wre(0, 1, k), xi = wim(0, 1, k);   // W1 * j^k
            for (int q = 0; q < (k & 3); ++q) { const double t = xr; xr = -xi; xi = t; }
            if (std::hypot(wre(0, 3, k) - xr, wim(0, 3, k) - xi) > 1e-13 * scale) tb->r13 = false;
        }
        sp.tables[i] = tb;
        sp.ok[i] = true;
    }
    return hipSuccess;
}

void systolic_release(SystolicPlan &sp) {
    for (int i = 0; i < 3; ++i) {
        if (sp.tables[i] && static_cast<Tables *>(sp.tables[i])->tw_dev) (void)hipFree(static_cast<Tables *>(sp.tables[i])->tw_dev);
        delete static_cast<Tables *>(sp.tables[i]);
        sp.tables[i] = nullptr;
        sp.ok[i] = false;
    }
    if (sp.sk_state) (void)hipFree(sp.sk_state);
    if (sp.sk_amp) (void)hipFree(sp.sk_amp);
    if (sp.thr_chunk) (void)hipFree(sp.thr_chunk);
    sp.sk_state = sp.sk_amp = sp.thr_chunk = nullptr;
    sp.sk_state_cap = sp.sk_amp_cap = sp.thr_chunk_cap = 0;
}

bool systolic_supports(const SystolicPlan &sp, int wsel, int T) {
    return wsel >= 0 && wsel < 3 && sp.ok[wsel] && T >= 1;  // (any number of sweeps: more than SYSTOLIC_MAX_ITERS run as several launches)
}

const char *systolic_name(const SystolicPlan &sp) { return sp.name; }

namespace {

// Shapes and scratch pointers of one call.
struct Geom {
    int Tp, Kr, Kt, G, TpPad, NT, nwg;   // Kr: rounds of ROWL frames (the kernel's clock); Kt: groups of 64 frames (layout tiles)
    int n_full, nwg_rest;   // batches larger than the chip: n_full spectrograms with one workgroup each, then the rest
                            // (B - n_full, fewer than there are CUs) with nwg_rest workgroups each
    int cb, rb;             // bytes per stored complex value / magnitude (Store<H16>)
    char *state_w, *state_nyq;
    char *amp_w, *amp_nyq;
    unsigned *amax_bits, *progress;
    float *thr_min;
    int *err;
};

// dense threshold table of one launch of a schedule longer than MAX_ITERS sweeps (a synchronising hipMalloc when it grows)
hipError_t ensure_thr_chunk(SystolicPlan &sp, int B, int iters) {
    const size_t need = (size_t)B * MAX_ITERS * sizeof(float);
    if (iters <= MAX_ITERS || (sp.thr_chunk && sp.thr_chunk_cap >= need)) return hipSuccess;
    if (sp.thr_chunk) (void)hipFree(sp.thr_chunk);
    sp.thr_chunk = nullptr; sp.thr_chunk_cap = 0;
    hipError_t e = hipMalloc(&sp.thr_chunk, need);
    if (e == hipSuccess) sp.thr_chunk_cap = need;
    return e;
}

// (grows the plan's scratch if needed -- a synchronising hipMalloc; systolic_reserve() does that ahead of time)
hipError_t prepare(SystolicPlan &sp, int B, int T, int iters, Geom &g) {
    const int Q = sp.Q, F = sp.F;
    g.cb = sp.h16 ? Store<true>::CB : Store<false>::CB;
    g.rb = sp.h16 ? Store<true>::RB : Store<false>::RB;
    g.Tp = T + 2 * (Q - 1);
    // rounds of ROWL frames per pass; at least so many that a pass (G rows) is longer than the lag of the last sweep slot behind
    // the loader plus what the loader reads ahead: pass g+1 follows pass g on the same clock and reads the rows pass g's last
    // slot has written (only the short-frame builds, with their 14 / 24 slots and short rounds, ever need the padding)
    constexpr int KR_MIN = (NSLOTS * LAG + 96 + ROWP - 1) / ROWP;
    g.Kr = std::max((g.Tp + ROWL - 1) / ROWL, KR_MIN);
    g.Kt = (g.Tp + LANES - 1) / LANES;
    g.G = ROWP * g.Kr;
    g.TpPad = (g.Tp + 63) & ~63;
    g.NT = (SKEW * (LANES - 1) + (F - 1) + TILE - 1) / TILE;   // time tiles per round of 64 frames
    // scratch: state_w, state_nyq | amp_w, amp_nyq, amax, smallest threshold, progress counters, error flag
    const size_t n_w = (size_t)B * g.G * ROWL, n_n = (size_t)B * g.TpPad;
    const size_t need_s = (n_w + n_n) * g.cb;
    hipError_t e;
    // workgroups per spectrogram: as many as there are CUs to keep busy and passes to share out; every workgroup must
    // be resident (they wait for each other), which one workgroup per CU (the rings fill the LDS) and a grid no larger
    // than the CU count guarantee on a device this process has to itself; if they are not (a shared or partitioned
    // device), the hand-over times out and the call is re-run with one workgroup per spectrogram (run_kernel)
    int n_cu = 0, dev = 0;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    if ((e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
    const int n_pass_max = ((iters < MAX_ITERS ? iters : MAX_ITERS) + NSLOTS - 1) / NSLOTS;
    // each workgroup trails its producer by NSLOTS*LAG + 56 rows, and the first one starts its next pass G rows after
    // its previous one: the lags around the ring must fit into one pass
    const int ring_max = g.G / (NSLOTS * LAG + 96);
    const char *ev = getenv("LWS_SYSTOLIC_NWG");
    auto pick = [&](int nb) {
        int n = nb > 0 ? n_cu / nb : 1;
        if (n > n_pass_max) n = n_pass_max;
        if (ev) { const int v = atoi(ev); if (v >= 1 && (long)v * nb <= n_cu) n = v; }
        if (n > ring_max) n = ring_max;
        return n < 1 ? 1 : n;
    };
    const int nwg = pick(B);
    g.nwg = nwg;
    // a batch that does not fill the last round of workgroups: the spectrograms of that round share the idle CUs
    g.n_full = B; g.nwg_rest = 1;
    if (nwg == 1 && B > n_cu && B % n_cu != 0 && pick(B % n_cu) > 1) { g.n_full = B - B % n_cu; g.nwg_rest = pick(B % n_cu); }
    // sized for the largest number of workgroups any iteration count can give this batch, so that the scratch of a shape
    // does not grow with the schedule
    const size_t n_prog = ((size_t)B * (size_t)(n_cu / (B > 0 ? B : 1) + 1) + (size_t)n_cu) * WPS + 8;
    const size_t amp_bytes = ((n_w + n_n) * g.rb + 15) & ~(size_t)15;
    const size_t need_a = amp_bytes + (size_t)B * sizeof(unsigned) + (size_t)B * sizeof(float) + n_prog * sizeof(unsigned);
    if (need_s > sp.sk_state_cap) {
        if (sp.sk_state) (void)hipFree(sp.sk_state);
        sp.sk_state = nullptr; sp.sk_state_cap = 0;
        if ((e = hipMalloc(&sp.sk_state, need_s)) != hipSuccess) return e;
        sp.sk_state_cap = need_s;
    }
    if (need_a > sp.sk_amp_cap) {
        if (sp.sk_amp) (void)hipFree(sp.sk_amp);
        sp.sk_amp = nullptr; sp.sk_amp_cap = 0;
        if ((e = hipMalloc(&sp.sk_amp, need_a)) != hipSuccess) return e;
        sp.sk_amp_cap = need_a;
    }
    g.state_w = static_cast<char *>(sp.sk_state);
    g.state_nyq = g.state_w + n_w * g.cb;
    g.amp_w = static_cast<char *>(sp.sk_amp);
    g.amp_nyq = g.amp_w + n_w * g.rb;
    g.amax_bits = reinterpret_cast<unsigned *>(g.amp_w + amp_bytes);
    g.thr_min = reinterpret_cast<float *>(g.amax_bits + B);
    g.progress = reinterpret_cast<unsigned *>(g.thr_min + B);
    g.err = reinterpret_cast<int *>(g.progress + (n_prog - 1));
    return hipSuccess;
}

hipError_t clear_flags(const Geom &g, int B, hipStream_t stream) {   // amax | thr_min | progress counters | error flag
    return hipMemsetAsync(g.amax_bits, 0, reinterpret_cast<char *>(g.err + 1) - reinterpret_cast<char *>(g.amax_bits), stream);
}

// One launch of the update kernel over spectrograms [b0, b0 + nb) with `nwg` workgroups each: n_it sweeps.
hipError_t launch_update(SystolicPlan &sp, const Geom &g, int wsel, const float *thr, int n_it, int b0, int nb, int nwg,
                         unsigned *progress, const int *gate, int T, hipStream_t stream) {
    const Tables *tb = static_cast<const Tables *>(sp.tables[wsel]);
    const int Q = sp.Q, L = sp.Lk, F = sp.F;
    SysArgs a;
    a.state_w = g.state_w + (size_t)b0 * g.G * ROWL * g.cb; a.amp_w = g.amp_w + (size_t)b0 * g.G * ROWL * g.rb;
    a.state_nyq = g.state_nyq + (size_t)b0 * g.TpPad * g.cb; a.amp_nyq = g.amp_nyq + (size_t)b0 * g.TpPad * g.rb;
    a.thr = thr + (size_t)b0 * n_it; a.amax = reinterpret_cast<const float *>(g.amax_bits) + b0;   // thr: dense [B][n_it]
    a.n_iters = n_it;
    a.T = T; a.Tp = g.Tp; a.TpPad = g.TpPad; a.Kr = g.Kr; a.G = g.G; a.C = F - 1;
    a.nwg = nwg; a.progress = progress; a.err = g.err; a.gate = gate;
    {
        const char *ev = getenv("LWS_SYSTOLIC_SPIN_LIMIT");   // polls (of ~0.2 us) before a workgroup gives up on its producer
        a.spin_limit = ev ? atoi(ev) : (1 << 21);
        const char *es = getenv("LWS_SYSTOLIC_STRESS");
        a.stress = es ? atoi(es) : 0;
        const char *er = getenv("LWS_SYSTOLIC_ROLEMAP");
        a.rolemap = er ? atoi(er) : 0;
    }
    a.tw_table = tb->tw_dev; a.tw_P = tb->tw_P > 0 ? tb->tw_P : 1; a.tw_invP = 1.0f / (float)a.tw_P;
    for (int r = 0; r < 8; ++r) {
        unsigned ur, ui;
        memcpy(&ur, &tb->tw_nyq[2 * r], 4); memcpy(&ui, &tb->tw_nyq[2 * r + 1], 4);
        a.tw_nyq[r] = ((unsigned long long)ui << 32) | ur;
    }
    a.Qa = Q;
    for (int x = 0; x < NW; ++x) {
        const bool used = LWS_Q8 || x < Q * (L + 1);
        const float re = used ? tb->w[2 * x] : 0.f, im = used ? tb->w[2 * x + 1] : 0.f;
        unsigned ur, ui;
        memcpy(&ur, &re, 4); memcpy(&ui, &im, 4);
        a.w[x] = ((unsigned long long)ui << 32) | ur;
    }
    const int grid = nb * nwg;
    const bool h = sp.h16;
    hipError_t e;
    const char *kind = "allmask";
#if LWS_TW && LWS_Q8
    // the Q = 8 kernel with the frame pairs r >= Q of the plan masked out (never fetched, never summed)
    kind = "tw";
#if LWS_TWQ
    if (Q != LWS_TWQ) return hipErrorInvalidValue;       // (systolic_build refuses every other Q)
    e = launch_k<8, 5, mask_all(LWS_TWQ, 5)>(a, grid, h, stream);
#else
    if (Q == 5) e = launch_k<8, 5, mask_all(5, 5)>(a, grid, h, stream);
    else if (Q == 6) e = launch_k<8, 5, mask_all(6, 5)>(a, grid, h, stream);
    else if (Q == 7) e = launch_k<8, 5, mask_all(7, 5)>(a, grid, h, stream);
    else e = launch_k<8, 5, mask_all(8, 5)>(a, grid, h, stream);
#endif
#elif LWS_TW
    kind = "tw";
    // hop = a third of the frame with the default sqrt-Hann window: of the centre frame's weights only k = 1 is non-zero (as for Q = 2, 4)
    constexpr uint64_t MASK_Q3_L5_DEFAULT = 0b111111'111111'000011u;
    if (Q == 4) e = launch_k<4, 5, mask_all(4, 5)>(a, grid, h, stream);
    else if (Q == 2) e = launch_k<2, 5, mask_all(2, 5)>(a, grid, h, stream);
    else if (tb->mask == MASK_Q3_L5_DEFAULT) { e = launch_k<3, 5, MASK_Q3_L5_DEFAULT>(a, grid, h, stream); kind = "hannmask_tw"; }
    else e = launch_k<3, 5, mask_all(3, 5)>(a, grid, h, stream);
#elif LWS_Q8
    // default sqrt-Hann weights at hop = window / 8: r = 0: {0,1}, r = 4: {0,1,2,4}, every other row all six taps
    constexpr uint64_t MASK_Q8_L5_DEFAULT = 0b111111'111111'111111'010111'111111'111111'111111'000011ull;
    if (tb->mask == MASK_Q8_L5_DEFAULT && tb->k0real) { e = launch_k<8, 5, MASK_Q8_L5_DEFAULT | FLAG_K0REAL>(a, grid, h, stream); kind = "hann"; }
    else e = launch_k<8, 5, mask_all(8, 5)>(a, grid, h, stream);
#elif LWS_R16
    if (L == 3) e = launch_k<2, 3, mask_all(2, 3)>(a, grid, h, stream);
    else if (L == 1) e = launch_k<2, 1, mask_all(2, 1)>(a, grid, h, stream);
    else if (tb->mask == MASK_Q2_L5_DEFAULT && tb->k0real) { e = launch_k<2, 5, MASK_Q2_L5_DEFAULT | FLAG_K0REAL>(a, grid, h, stream); kind = "hann"; }
    else if (tb->mask == MASK_Q2_L5_DEFAULT) { e = launch_k<2, 5, MASK_Q2_L5_DEFAULT>(a, grid, h, stream); kind = "hannmask"; }
    else e = launch_k<2, 5, mask_all(2, 5)>(a, grid, h, stream);
#elif LWS_L7
    e = Q == 4 ? launch_k<4, 7, mask_all(4, 7)>(a, grid, h, stream) : launch_k<2, 7, mask_all(2, 7)>(a, grid, h, stream);
#else
    if (L == 3) {
        e = Q == 4 ? launch_k<4, 3, mask_all(4, 3)>(a, grid, h, stream) : launch_k<2, 3, mask_all(2, 3)>(a, grid, h, stream);
    } else if (L == 1) {
        e = Q == 4 ? launch_k<4, 1, mask_all(4, 1)>(a, grid, h, stream) : launch_k<2, 1, mask_all(2, 1)>(a, grid, h, stream);
    } else if (Q == 4) {
        if (tb->mask == MASK_Q4_L5_DEFAULT && tb->k0real && tb->r13) { e = launch_k<4, 5, MASK_Q4_L5_DEFAULT | FLAG_K0REAL | FLAG_R13>(a, grid, h, stream); kind = "hann"; }
        else if (tb->mask == MASK_Q4_L5_DEFAULT) { e = launch_k<4, 5, MASK_Q4_L5_DEFAULT>(a, grid, h, stream); kind = "hannmask"; }
        else e = launch_k<4, 5, mask_all(4, 5)>(a, grid, h, stream);
    } else {
        if (tb->mask == MASK_Q2_L5_DEFAULT && tb->k0real) { e = launch_k<2, 5, MASK_Q2_L5_DEFAULT | FLAG_K0REAL>(a, grid, h, stream); kind = "hann"; }
        else if (tb->mask == MASK_Q2_L5_DEFAULT) { e = launch_k<2, 5, MASK_Q2_L5_DEFAULT>(a, grid, h, stream); kind = "hannmask"; }
        else e = launch_k<2, 5, mask_all(2, 5)>(a, grid, h, stream);
    }
#endif
    snprintf(sp.name_buf, sizeof sp.name_buf, "systolic%s_q%d_l%d_%s%s", (LWS_TW && SPW == 2) ? "_half" : (LWS_TW && LWS_WIDE) ? "_wide" : (LWS_TWQ == 5) ? "_r40" : (LWS_TWQ == 6) ? "_r48" : (LWS_TW && LWS_Q8) ? "_r64" : LWS_TW ? "" : (LWS_R16 && LWS_WIDE) ? "_wide_r16" : (LWS_R16 && SPW == 2) ? "_half_r16" : (LWS_R16 && SPW == 4) ? "_quarter_r16" : LWS_R16 ? "_r16" : LWS_WIDE == 2 ? "_xwide" : LWS_WIDE ? "_wide" : (SPW == 2 ? "_half" : (SPW == 4 ? "_quarter" : "")), Q, L, kind,
             h ? "_f16" : "");
    sp.name = sp.name_buf;
    return e;
}

// All sweeps of a call, state in the skewed layout.  `reload(gate)` re-creates that layout from the caller's (still
// untouched) data, as a launch that only runs if *gate != 0.
template <typename Reload>
hipError_t run_kernel(SystolicPlan &sp, const Geom &g, int wsel, const float *thr, int B, int T, int iters,
                      hipStream_t stream, int *launches, Reload reload) {
    hipError_t e = hipSuccess;
    if (sp.h16) {
        hipLaunchKernelGGL(k_thr_min, dim3(B), dim3(64), 0, stream, thr, reinterpret_cast<const float *>(g.amax_bits), iters, g.thr_min);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if ((e = ensure_thr_chunk(sp, B, iters)) != hipSuccess) return e;
    int n_launch = 0;
    bool multi = false, refused0 = false, refused1 = false;
    // every sweep of the call, either with the workgroup counts of `g` or (single) with one workgroup per spectrogram
    auto sweep_all = [&](bool single, const int *gate) -> hipError_t {
        // more sweeps than one launch's threshold table holds: several launches over the same (resident) skewed state.
        // Same results: a launch boundary is just a longer lag between two sweeps.
        for (int it0 = 0; it0 < iters; it0 += MAX_ITERS) {
            const int n_it = iters - it0 < MAX_ITERS ? iters - it0 : MAX_ITERS;
            const float *thr_c = thr;
            if (iters > MAX_ITERS) {   // the kernel reads a dense [B][n_iters] table: this launch's columns, copied in stream order
                if ((e = hipMemcpy2DAsync(sp.thr_chunk, (size_t)n_it * sizeof(float), thr + it0, (size_t)iters * sizeof(float),
                                          (size_t)n_it * sizeof(float), B, hipMemcpyDeviceToDevice, stream)) != hipSuccess) return e;
                thr_c = static_cast<const float *>(sp.thr_chunk);
            }
            for (int chunk = 0; chunk < (single ? 1 : 2); ++chunk) {
                // chunk 0: the first n_full spectrograms (all of them unless the batch leaves a partial last round on the
                // chip); chunk 1: the rest, with several workgroups per spectrogram
                const int b0 = (single || chunk == 0) ? 0 : g.n_full;
                const int nb = single ? B : (chunk == 0 ? g.n_full : B - g.n_full);
                const int nwg = single ? 1 : (chunk == 0 ? g.nwg : g.nwg_rest);
                if (nb <= 0) continue;
                unsigned *progress = g.progress + (chunk == 0 ? 0 : (size_t)B * g.nwg * WPS);
                if (nwg > 1 && n_launch > 0) {   // multi-workgroup launches re-use the counters: start them from zero again
                    if ((e = hipMemsetAsync(progress, 0, (size_t)nb * nwg * WPS * sizeof(unsigned), stream)) != hipSuccess) return e;
                }
                e = launch_update(sp, g, wsel, thr_c, n_it, b0, nb, nwg, progress, gate, T, stream);
                if (e == hipErrorLaunchOutOfResources && nwg > 1 && it0 == 0) {
                    // the occupancy guard refused a grid of several workgroups per spectrogram (launch_km): nothing was launched for
                    // these spectrograms -- serve them with one workgroup each (same results), here and in the launches that follow
                    (void)hipGetLastError();
                    if (chunk == 0) refused0 = true; else refused1 = true;
                }
                if ((chunk == 0 ? refused0 : refused1) && nwg > 1)
                    e = launch_update(sp, g, wsel, thr_c, n_it, b0, nb, 1, progress, gate, T, stream);
                else multi |= nwg > 1;
                if (e != hipSuccess) return e;
                if (!gate) ++n_launch;
            }
        }
        return hipSuccess;
    };
    if ((e = sweep_all(false, nullptr)) != hipSuccess) return e;
    if (multi) {
        // Several workgroups per spectrogram hand rows over through HBM and need each other resident.  If one of them
        // timed out (flag set), everything is done again with one workgroup per spectrogram, from the caller's data: the
        // launches below are enqueued unconditionally and return at once unless the flag is set, so the call stays
        // asynchronous and never returns the results of a failed hand-over.
        if ((e = reload(g.err)) != hipSuccess) return e;
        if ((e = sweep_all(true, g.err)) != hipSuccess) return e;
    }
    sp.err_dev = g.err; sp.last_nwg = multi ? (g.nwg > g.nwg_rest ? g.nwg : g.nwg_rest) : 1;
    if (launches) *launches = n_launch;
    return e;
}

}  // namespace

hipError_t systolic_reserve(SystolicPlan &sp, int B, int T, int iters) {
    Geom g;
    hipError_t e = prepare(sp, B, T, iters, g);
    if (e != hipSuccess) return e;
    return ensure_thr_chunk(sp, B, iters);   // (schedules longer than one launch's table: sized here, not in the first call)
}

hipError_t launch_systolic(SystolicPlan &sp, int wsel, float2 *state, const float *amp, const float *thr, int B,
                           int T, int iters, hipStream_t stream, int *launches, hipEvent_t ev0, hipEvent_t ev1) {
    Geom g;
    hipError_t e;
    if ((e = prepare(sp, B, T, iters, g)) != hipSuccess) return e;
    if ((e = clear_flags(g, B, stream)) != hipSuccess) return e;
    const int Q = sp.Q, L = sp.L, F = sp.F;
    auto load = [&](const int *gate) {
        if (sp.h16) {
            if (!gate) hipLaunchKernelGGL(k_amax_ext, dim3(64, B), dim3(256), 0, stream, amp, g.amax_bits, T, F + 2 * L, Q);
            hipLaunchKernelGGL(k_to_skew<true>, dim3(g.Kt * g.NT, B), dim3(256), 0, stream, state, amp, (void *)g.state_w, (void *)g.amp_w,
                               (void *)g.state_nyq, (void *)g.amp_nyq, g.amax_bits, T, F, L, Q, g.G, g.TpPad, g.NT, gate);
        } else {
            hipLaunchKernelGGL(k_to_skew<false>, dim3(g.Kt * g.NT, B), dim3(256), 0, stream, state, amp, (void *)g.state_w, (void *)g.amp_w,
                               (void *)g.state_nyq, (void *)g.amp_nyq, g.amax_bits, T, F, L, Q, g.G, g.TpPad, g.NT, gate);
        }
        return hipGetLastError();
    };
    if ((e = load(nullptr)) != hipSuccess) return e;
    if (ev0) (void)hipEventRecord(ev0, stream);
    if ((e = run_kernel(sp, g, wsel, thr, B, T, iters, stream, launches, load)) != hipSuccess) return e;
    if (ev1) (void)hipEventRecord(ev1, stream);
    const float *amax = reinterpret_cast<const float *>(g.amax_bits);
    if (sp.h16)
        hipLaunchKernelGGL(k_from_skew<true>, dim3(g.Kt * g.NT, B), dim3(256), 0, stream, state, amp, (const void *)g.state_w,
                           (const void *)g.state_nyq, amax, (const float *)g.thr_min, T, F, L, Q, g.G, g.TpPad, g.NT);
    else
        hipLaunchKernelGGL(k_from_skew<false>, dim3(g.Kt * g.NT, B), dim3(256), 0, stream, state, amp, (const void *)g.state_w,
                           (const void *)g.state_nyq, amax, (const float *)g.thr_min, T, F, L, Q, g.G, g.TpPad, g.NT);
    return hipGetLastError();
}

// ---- a call that is one batch stage on the caller's unpadded complex64 spectrograms: no extended buffers at all
namespace {
size_t io_partials_n(int F, int T, int Q) {            // one partial sum per tile of k_in_to_skew
    const int NT = (SKEW * (LANES - 1) + (F - 1) + TILE - 1) / TILE;
    const int Tp = T + 2 * (Q - 1), Kt = (Tp + LANES - 1) / LANES;
    return (size_t)Kt * NT;
}
hipError_t io_load(SystolicPlan &sp, const Geom &g, const float2 *in, int B, int T, double *partial, hipStream_t stream, const int *gate) {
    if (sp.h16) {
        if (!gate) hipLaunchKernelGGL(k_amax_in, dim3(64, B), dim3(256), 0, stream, in, g.amax_bits, (size_t)T * sp.F);
        hipLaunchKernelGGL(k_in_to_skew<true>, dim3(g.Kt * g.NT, B), dim3(256), 0, stream, in, (void *)g.state_w, (void *)g.amp_w,
                           (void *)g.state_nyq, (void *)g.amp_nyq, g.amax_bits, partial, T, sp.F, sp.Q, g.G, g.TpPad, g.NT, gate);
    } else {
        hipLaunchKernelGGL(k_in_to_skew<false>, dim3(g.Kt * g.NT, B), dim3(256), 0, stream, in, (void *)g.state_w, (void *)g.amp_w,
                           (void *)g.state_nyq, (void *)g.amp_nyq, g.amax_bits, partial, T, sp.F, sp.Q, g.G, g.TpPad, g.NT, gate);
    }
    return hipGetLastError();
}
}  // namespace
size_t systolic_io_partials(const SystolicPlan &sp, int T) { return io_partials_n(sp.F, T, sp.Q); }

hipError_t systolic_io_load(SystolicPlan &sp, const float2 *in, int B, int T, int iters, double *partial, double *mean_amp,
                            hipStream_t stream) {
    Geom g;
    hipError_t e;
    if ((e = prepare(sp, B, T, iters, g)) != hipSuccess) return e;
    if ((e = clear_flags(g, B, stream)) != hipSuccess) return e;
    const size_t n = io_partials_n(sp.F, T, sp.Q);
    if ((e = io_load(sp, g, in, B, T, partial, stream, nullptr)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_mean_partials, dim3(B), dim3(256), 0, stream, partial, mean_amp, (int)n, (double)T * (double)sp.F);
    return hipGetLastError();
}

hipError_t systolic_io_run(SystolicPlan &sp, int wsel, const float *thr, const float2 *in, float2 *out, double *partial, int B, int T,
                           int iters, hipStream_t stream, int *launches, hipEvent_t ev0, hipEvent_t ev1) {
    Geom g;
    hipError_t e;
    if ((e = prepare(sp, B, T, iters, g)) != hipSuccess) return e;   // same shapes: no reallocation, same pointers
    if (ev0) (void)hipEventRecord(ev0, stream);
    auto reload = [&](const int *gate) { return io_load(sp, g, in, B, T, partial, stream, gate); };
    if ((e = run_kernel(sp, g, wsel, thr, B, T, iters, stream, launches, reload)) != hipSuccess) return e;
    if (ev1) (void)hipEventRecord(ev1, stream);
    const float *amax = reinterpret_cast<const float *>(g.amax_bits);
    if (sp.h16)
        hipLaunchKernelGGL(k_skew_to_out<true>, dim3(g.Kt * g.NT, B), dim3(256), 0, stream, out, in, (const void *)g.state_w,
                           (const void *)g.state_nyq, amax, (const float *)g.thr_min, T, sp.F, sp.Q, g.G, g.TpPad, g.NT);
    else
        hipLaunchKernelGGL(k_skew_to_out<false>, dim3(g.Kt * g.NT, B), dim3(256), 0, stream, out, in, (const void *)g.state_w,
                           (const void *)g.state_nyq, amax, (const float *)g.thr_min, T, sp.F, sp.Q, g.G, g.TpPad, g.NT);
    return hipGetLastError();
}

const SystolicBuild &systolic_entry() {
    static const SystolicBuild b = {systolic_build, systolic_release, systolic_supports, systolic_reserve, systolic_name,
                                    launch_systolic, systolic_io_partials, systolic_io_load, systolic_io_run};
    return b;
}

LWS_NS_CLOSE
