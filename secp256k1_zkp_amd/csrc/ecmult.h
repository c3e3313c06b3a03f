// ecmult.h -- per-lane double multiplication  R = na*A + ng*G  (the reference's secp256k1_ecmult,
// src/ecmult_impl.h:365-375 -> strauss_wnaf :252-363) re-designed for a 64-wide SIMT machine.
//
// The reference interleaves wNAF(5) of the GLV halves of na with wNAF(15) of ng and walks 129 doublings; its per-call
// table (8 odd multiples + lambda copies) lives on the CPU stack.  On gfx950 a sparse wNAF wastes lanes -- an add slot
// costs the whole wavefront whenever ANY lane has a digit -- so every slot is made useful for every lane:
//
//   variable point:  GLV split (scalar.h); each 129-bit half is recoded into 33 signed ODD 4-bit digits (never zero;
//                    digit i is read straight off the bits of the scalar, no carry chain), so the loop is exactly
//                    4 doublings + 2 additions per digit position for all 64 lanes.  The 8 odd multiples {1..15}P
//                    (+ their beta*x for lambda*P) are built once per multiplication with one common Z -- the reference's
//                    isomorphic-curve / global-Z trick (ecmult_impl.h:73-115, group_impl.h:289-320) -- and parked in a
//                    per-lane 1152-byte slice of HBM (L2/MALL resident while the lane is alive); operands are gathered one
//                    addition ahead so the ~1-2 us of latency hides under the previous ~4 us of arithmetic.
//   generator:       no doublings at all: ng is cut into 10 signed digits of 26 bits (S2K_GTAB_BITS), each indexing a precomputed
//                    (window, |digit|) -> affine multiple table (gtable.h, 21.5 GB of the 288 GB of HBM), 10 mixed additions.
//   digits:          both digit streams live in LDS (lane_mem), not in registers.
//   control:         one loop whose body contains exactly ONE doubling site and ONE mixed-add site, driven by a
//                    per-lane micro-program counter.  The only data-dependent *arithmetic* case (P + P inside an add)
//                    becomes "take the operand and double it on the next trip", so exceptional inputs cost one extra
//                    trip for that lane instead of a second copy of the doubling code.
//
// Results are identical to the reference as group elements (and therefore as serialised bytes).
#pragma once
#include "group.h"
#include "scalar.h"

// ---- generator table ---------------------------------------------------------------------------------
// Fixed-base table of G (and, same layout, of a rangeproof generator): SIGNED digits of D = S2K_GTAB_BITS bits.  A scalar s < 2^256 is
// cut as  s = sum_w d_w 2^(D w)  with d_w in [-2^(D-1), 2^(D-1)) for every window but the top one (which takes what is left, >= 0):
// adding the constant K = sum_{w < W-1} 2^(D-1 + D w) once turns that into plain unsigned windows t_w of s' = s + K with d_w = t_w - 2^(D-1)
// (gtab_recode; s' has up to 257 bits: 9 words), so a window is still read straight off the stored words, with no carry chain.  The
// table holds v * 2^(D w) * G for v = 1 .. 2^(D-1) only -- a negative digit negates y when the operand is decoded -- i.e. HALF the entries
// an unsigned window of the same width needs: D = 26 gives W = 10 windows (one addition less per multiplication than the 11 unsigned
// 24-bit windows of rounds 1-3) in 10 x 2^25 x 64 B = 21.5 GB of the 288 GB; D = 24 would be 11 windows in 5.9 GB.
// Slot (w, v) = (w << (D-1)) + v: v = 2^(D-1) lands on slot (w+1, 0), which no window ever reads (a zero digit adds nothing).
// One entry = one aligned 64-byte sector of canonical words (x[8], y[8], least significant first), the same record format as the
// per-lane tables below, so that the main loops have a single operand pipeline.  (The host emulation builds D = 12.)
// Round 5: the width D is a property of the TABLE, not of the code -- the engine builds D = 26 where 21.5 GB are to be had and falls back to
// narrower tables where they are not (24: 5.9 GB, 22: 1.6 GB, 20: 0.44 GB; one more addition per multiplication for every step down; the
// reference's knob of the same kind is ECMULT_WINDOW_SIZE, src/ecmult.h:14-38).  Slot (0, 0) -- the only slot no digit ever addresses and
// no entry is stored in -- is the table's header: word 0 = D, word 1 = the number of windows, words 2..10 = the recoding constant K.
// Every routine below takes its geometry from there (one uniform load per multiplication); S2K_GTAB_BITS is only the DEFAULT width.
#ifndef S2K_GTAB_BITS
#define S2K_GTAB_BITS 26
#endif
#define S2K_GTAB_MAX_BITS 26
#define S2K_GTAB_ENTRY_WORDS 16
#define S2K_GTAB_SWORDS 9           /* words of a recoded scalar */
#define S2K_GTAB_MAX_WINDOWS 32     /* D >= 8 */
struct gtab_geom { u32 D, W; };     // digit width, number of windows
S2K_HD u32 gtab_windows_for(u32 D) { return (256u + D - 1u) / D; }
S2K_HD u32 gtab_top_bits_for(u32 D) { return 256u - D * (gtab_windows_for(D) - 1u); }      // the top window's value is at most 2^TOP_BITS (its bits + the carry)
S2K_HD int gtab_bits_ok(u32 D) { return D >= 8u && D <= S2K_GTAB_MAX_BITS && gtab_top_bits_for(D) >= 1u && gtab_top_bits_for(D) < D - 1u; }      // the top window must fit the table
S2K_HD size_t gtab_slot(u32 D, u32 w, u32 v) { return ((size_t)w << (D - 1u)) + (size_t)v; }
S2K_HD size_t gtab_slots_for(u32 D) { return gtab_slot(D, gtab_windows_for(D), 0) + 1; }
S2K_HD size_t gtab_words_for(u32 D) { return gtab_slots_for(D) * S2K_GTAB_ENTRY_WORDS; }
S2K_HD gtab_geom gtab_geometry(const u32* tab) { gtab_geom g; g.D = tab[0]; g.W = tab[1]; return g; }
// the header of a table of width D (written once, by the kernel that builds the window bases)
S2K_HD void gtab_write_header(u32* tab, u32 D) {
    const u32 W = gtab_windows_for(D);
    tab[0] = D; tab[1] = W;
    for (int i = 0; i < S2K_GTAB_SWORDS; i++) {
        u32 k = 0;
        for (u32 w = 0; w + 1 < W; w++) { const u32 bit = D - 1u + D * w; if ((bit >> 5) == (u32)i) k |= 1u << (bit & 31u); }
        tab[2 + i] = k;
    }
    for (int i = 2 + S2K_GTAB_SWORDS; i < S2K_GTAB_ENTRY_WORDS; i++) tab[i] = 0;
}

S2K_HD void gtab_load(ge& r, const u32* gtab, u32 window, u32 v) {
    const u32* p = gtab + gtab_slot(gtab[0], window, v) * S2K_GTAB_ENTRY_WORDS;
    u32 w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = p[i];
    fe_from_words(r.x, w); fe_from_words(r.y, w + 8);
}
// s (8 little-endian words) -> s' = s + K (9 words), K from the header of the table the digits will index
S2K_HD void gtab_recode(u32 out[S2K_GTAB_SWORDS], const u32 s[8], const u32* tab) {
    u64 cy = 0;
#pragma unroll
    for (int i = 0; i < S2K_GTAB_SWORDS; i++) {
        cy += (u64)(i < 8 ? s[i] : 0u) + tab[2 + i];
        out[i] = (u32)cy; cy >>= 32;
    }
}
// window g of a recoded scalar, given the two words that hold its bits (lo = word (D g) >> 5, hi = the next one or 0): the table record to
// add (returns 0: the digit is zero) and whether its y has to be negated
S2K_HD int gtab_locate(const u32*& addr, int& neg, const u32* tab, const gtab_geom& G, int g, u32 lo, u32 hi) {
    const u32 b = (u32)g * G.D;
    const u64 pair = (u64)lo | ((u64)hi << 32);
    const u32 t = (u32)(pair >> (b & 31u)) & ((1u << G.D) - 1u);
    u32 v;
    if ((u32)g + 1u < G.W) { const int d = (int)t - (int)(1u << (G.D - 1u)); neg = d < 0; v = (u32)(d < 0 ? -d : d); }
    else { neg = 0; v = t; }
    if (!v) return 0;
    addr = tab + gtab_slot(G.D, (u32)g, v) * S2K_GTAB_ENTRY_WORDS;
    return 1;
}

// ---- per-lane table of odd multiples ----------------------------------------------------------------------
// ptab (this lane's slice of HBM, 128-byte aligned): entry e = 0..7 holds (2e+1)*P on the isomorphic curve as two
// 64-byte sectors of canonical 8x32-bit words,  [ x | y ]  and  [ beta*x | y ] , so that an operand for either GLV half
// is exactly ONE aligned 64-byte gather (the per-lane tables of all resident waves live in the Infinity Cache, not L2:
// every gather crosses the fabric, so sectors are what is paid for).  During construction the same 128 bytes hold the
// 27 limbs (x, y, z-ratio) of the not-yet-rescaled entry.
// S2K_PTAB_TWINS = 0 (default): only [ x | y ] is kept and an operand for the lambda half costs one multiplication by beta when it
// is decoded.  The finished tables of a lane are then 16 sectors = 1 KB (two-piece form) instead of 2 KB, i.e. 134 MB instead of
// 268 MB for the 131 072 resident lanes of a rangeproof launch -- inside the 256 MB Infinity Cache instead of just beyond it -- and
// the rescaling pass saves one product, one normalisation and half of its stores per entry.  S2K_PTAB_TWINS = 1 keeps [ beta*x | y ].
#ifndef S2K_PTAB_TWINS
#define S2K_PTAB_TWINS 0
#endif
#define S2K_PTAB_ENTRIES 8
#define S2K_PTAB_ENTRY_WORDS 32
#define S2K_PTAB_TABLE_WORDS (S2K_PTAB_ENTRIES * S2K_PTAB_ENTRY_WORDS)
#define S2K_PTAB_ZISO (2 * S2K_PTAB_TABLE_WORDS)          // parked while the main loop runs, to keep VGPRs for arithmetic
#define S2K_PTAB_NG (S2K_PTAB_ZISO + 9)
#define S2K_PTAB_WORDS (2 * S2K_PTAB_TABLE_WORDS + 32)    // two tables (the second one only in the split form below) + parked factors

// A parked entry is 27 words (x, y, third); WS = distance between its consecutive words: 1 for an entry that lies in one piece in the
// lane's own slice, 64 for the wave-interleaved parking area of the ring form (word k of the 64 lanes of a wavefront side by side, so
// that one store / load instruction moves 256 contiguous bytes instead of touching 64 different lines).
// Non-temporal hints for data that is touched once, so that it does not displace the finished per-lane tables (2 KB per lane, re-read ~50
// times per ring position, 268 MB over the resident lanes: right at the size of the Infinity Cache) from the caches.  S2K_NT_PARK: the parked
// entries of a table under construction in the ring form's SEPARATE, wave-interleaved parking area (written once, read once by the rescaling
// pass; NOT the in-place parking of the general form, whose lines the finished sectors overwrite at once: hinted, it lost 15 %); S2K_NT_GTAB: the operands the shared-generator
// ring form takes from the 21.5 GB fixed-base tables (random sectors, never reused).  Measured together on one box (tools/ab_probe.py,
// profiles/r04v_ab_nontemporal.txt): 1.0425e6 -> 1.053e6 verifies/s; either one alone is inside the noise (+-0.4 %).  0 = plain accesses.
#if defined(__HIP_DEVICE_COMPILE__)
#define S2K_LD_NT(p) __builtin_nontemporal_load(p)
#define S2K_ST_NT(v, p) __builtin_nontemporal_store((v), (p))
#else
#define S2K_LD_NT(p) (*(p))
#define S2K_ST_NT(v, p) (*(p) = (v))
#endif
#ifndef S2K_NT_PARK
#define S2K_NT_PARK 1
#endif
// S2K_XYZZ_TABLE_PART: the run of fixed-base additions at the end of a multiplication (W from G's table, W more from the generator's in the
// ring form) has no doubling in between, so its accumulator can stay in extended Jacobian form (X, Y, ZZ, ZZZ: group.h): an addition then
// costs 8M + 2S instead of 8M + 3S, and the same-x test of every addition becomes ONE zero test of ZZ behind the run (an exceptional
// addition zeroes ZZ for good; additions of zero digits are not committed, so they cannot).  In: ZZ = Z^2, ZZZ = Z ZZ; out: (X ZZ, Y ZZZ, ZZ).
// A/B on one box, three alternating rounds (profiles/r05x_ab_xyzz_table_part.txt): general form (every proof its own generator) +0.9 % in
// every round, shared-generator form +0.3 % (inside its noise), BIP-340 (ecmult_lane: its table part is not a loop of its own) unchanged.
#ifndef S2K_XYZZ_TABLE_PART
#define S2K_XYZZ_TABLE_PART 1
#endif
#ifndef S2K_NT_GTAB_SPLIT
#define S2K_NT_GTAB_SPLIT 0      /* the same hint on the generator part of ecmult_lane_split (general form of the rings kernel): A/B in profiles/r05*_ab_* */
#endif
#ifndef S2K_NT_GTAB
#define S2K_NT_GTAB 1
#endif
template <int WS = 1>
S2K_HD void ptab_store_raw(u32* e, const fe& x, const fe& y, const fe& third) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
        if (S2K_NT_PARK && WS > 1) { S2K_ST_NT(x.n[i], &e[i * WS]); S2K_ST_NT(y.n[i], &e[(9 + i) * WS]); S2K_ST_NT(third.n[i], &e[(18 + i) * WS]); }
        else { e[i * WS] = x.n[i]; e[(9 + i) * WS] = y.n[i]; e[(18 + i) * WS] = third.n[i]; }
    }
}
// Table construction in two passes.  ptab_build_raw: the odd multiples by co-Z additions of 2A, each entry parked as (x, y, z-ratio of
// its step) limbs; returns the Z all of them will share once rescaled (the Z of the last entry).  ptab_rescale: brings every entry to the Z of the last one times `zs0`
// (secp256k1_ge_table_set_globalz, group_impl.h:289-320) and packs it as canonical words with its beta*x twin.  zs0 = 1 for a
// single table; with two tables each is rescaled by the OTHER one's Z so that all sixteen entries share one Z (ecmult_lane_split).
template <int N, int WS = 1, int ES = S2K_PTAB_ENTRY_WORDS>
S2K_HD void ptab_build_raw_n(fe& ziso, u32* tab, const gej& A) {
    // 2A by the usual doubling, whose intermediates also give A itself at the Z of 2A for free: Z(2A) = Y Z, so A rescaled by Y is
    // (X Y^2, Y^4) = (-T, S^2).  From then on every odd multiple is a CO-Z addition of 2A (both operands share Z: 5M + 2S instead of the
    // 8M + 3S of a mixed addition, and 2A comes out rescaled to the sum's Z for the next step); the step's Z ratio is just X(2A) - X(P).
    fe x = A.x, y = A.y; fe_norm_weak(x); fe_norm_weak(y);
    fe zc, s, t, l, nx, s2, qx, qy, w, px, py;
    fe_mul_sqr(zc, y, A.z, s, y);                          // Z(2A) = Y Z, S = Y^2
    fe_neg(nx, x, 1);
    fe_mul_sqr(t, nx, s, l, x);                            // T = -X S, X^2
    fe_mul_int(l, 3); fe_half(l); fe_norm_weak(l);         // L = 3/2 X^2
    fe_sqr2(qx, l, s2, s);                                 // L^2, S^2
    fe_add(qx, t); fe_add(qx, t); fe_norm_weak(qx);        // X(2A)                                  (1)
    fe_add2(w, qx, t);
    fe_mul(qy, l, w); fe_add(qy, s2); fe_neg(qy, qy, 2); fe_norm_weak(qy);     // Y(2A) = -(L (X3 + T) + S^2)      (1)
    fe_neg(px, t, 1); py = s2;                             // A at the Z of 2A                      (2, 1)
    { fe one; fe_set_int(one, 1); ptab_store_raw<WS>(tab, px, py, one); }
    for (int i = 1; i < N; i++) {
        fe dx, dy, c, d, w1, w2, e, a1, x3, y3, tmp;
        fe_neg(dx, px, 4); fe_add(dx, qx); fe_norm_weak(dx);              // X(2A) - X(P): also Z(sum) / Z(operands)
        fe_neg(dy, py, 3); fe_add(dy, qy); fe_norm_weak(dy);
        fe_sqr2(c, dx, d, dy);
        fe_mul2(w1, qx, c, w2, px, c);                                    // (1*1, 4*1)
        fe_neg(e, w2, 1); fe_add(e, w1);                                  // W1 - W2                     (3)
        fe_add2(tmp, w1, w2); fe_neg(tmp, tmp, 2);                        // -(W1 + W2)                  (3)
        fe_add2(x3, d, tmp);                                              // X(P + 2A)                   (4)
        fe_neg(tmp, x3, 4); fe_add(tmp, w1);                              // W1 - X3                     (6)
        fe_mul2(a1, qy, e, y3, dy, tmp);                                  // (1*3, 1*6)
        fe_neg(tmp, a1, 1); fe_add(y3, tmp);                              // Y(P + 2A)                   (3)
        fe_mul(zc, zc, dx);
        ptab_store_raw<WS>(tab + i * ES, x3, y3, dx);                      // third slot: z ratio of this step
        px = x3; py = y3; qx = w1; qy = a1;                               // 2A rescaled to the new Z
    }
    ziso = zc;
}
S2K_HD void ptab_build_raw(fe& ziso, u32* tab, const gej& A) { ptab_build_raw_n<S2K_PTAB_ENTRIES>(ziso, tab, A); }
// N parked entries at `tab` (one per S2K_PTAB_ENTRY_WORDS slot) -> N finished 64-byte sectors at fin + i * fin_stride.  fin == tab with
// fin_stride == S2K_PTAB_ENTRY_WORDS is the in-place form (a finished sector overwrites the head of its own parked entry, which has
// been taken over into registers by then).
template <int N, int WS = 1, int ES = S2K_PTAB_ENTRY_WORDS>
S2K_HD void ptab_rescale_n(u32* tab, u32* fin, int fin_stride, const fe* zs0) {
    fe zs; if (zs0) zs = *zs0; else fe_set_int(zs, 1);
    // The parked entry of step i-1 is requested before the arithmetic of step i and taken over after it (the first use of the loaded
    // words is where the wait lands; x, y, h are a second register set, so no copy of in-flight data sits at the loop head).  Exactly
    // 27 words are loaded: a 28th, dead, destination register would be recycled by the allocator while the load is in flight, and
    // the write-after-write hazard then puts a full wait right behind the request.
    u32 nraw[27];
    fe x, y, h;
#pragma unroll
    for (int k = 0; k < 27; k++) nraw[k] = (S2K_NT_PARK && WS > 1) ? S2K_LD_NT(&tab[(N - 1) * ES + k * WS]) : tab[(N - 1) * ES + k * WS];
#pragma unroll
    for (int k = 0; k < 9; k++) { x.n[k] = nraw[k]; y.n[k] = nraw[9 + k]; h.n[k] = nraw[18 + k]; }
    for (int i = N - 1; i >= 0; i--) {
        const u32* er = tab + i * ES;
        u32* e = fin + i * fin_stride;
        if (i > 0) {
#pragma unroll
            for (int k = 0; k < 27; k++) nraw[k] = (S2K_NT_PARK && WS > 1) ? S2K_LD_NT(&er[k * WS - ES]) : er[k * WS - ES];
        }
        if (zs0 || i != N - 1) {
            fe zs2, zs3; fe_sqr(zs2, zs); fe_mul(zs3, zs2, zs);
            fe_mul(x, x, zs2); fe_mul(y, y, zs3);
        }
        fe_normalize(x); fe_normalize(y);
        u32 wx[8], wy[8];
        fe_to_words(wx, x); fe_to_words(wy, y);
#pragma unroll
        for (int k = 0; k < 8; k++) { e[k] = wx[k]; e[8 + k] = wy[k]; }
#if S2K_PTAB_TWINS
        {
            fe beta, bx; fe_set_beta(beta);
            fe_mul(bx, x, beta); fe_normalize(bx);
            u32 wb[8]; fe_to_words(wb, bx);
#pragma unroll
            for (int k = 0; k < 8; k++) { e[16 + k] = wb[k]; e[24 + k] = wy[k]; }
        }
#endif
        fe_mul(zs, zs, h);             // ratio z_i / z_{i-1} joins the running product for the entries below
        if (i > 0) {
#pragma unroll
            for (int k = 0; k < 27; k++) S2K_CHAIN(nraw[k]);
#pragma unroll
            for (int k = 0; k < 9; k++) { x.n[k] = nraw[k]; y.n[k] = nraw[9 + k]; h.n[k] = nraw[18 + k]; }
        }
    }
}
S2K_HD void ptab_rescale(u32* tab, const fe* zs0) { ptab_rescale_n<S2K_PTAB_ENTRIES>(tab, tab, S2K_PTAB_ENTRY_WORDS, zs0); }
// Builds the table for a finite Jacobian A (magnitudes <= (5,3,1)); returns the factor that takes the accumulator's Z
// from the table's common-Z frame back to the real curve.  ~105 field multiplications.
S2K_HD void ptab_build(fe& ziso, u32* ptab, const gej& A) {
    ptab_build_raw(ziso, ptab, A);
    ptab_rescale(ptab, nullptr);
}
S2K_HD void ptab_load_ziso(fe& zi, const u32* ptab) {
#pragma unroll
    for (int i = 0; i < 9; i++) zi.n[i] = ptab[S2K_PTAB_ZISO + i];
}

// ---- wave-level predicates ------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
#define S2K_WAVE_ANY(p) (__any(p))
#define S2K_WAVE_ALL(p) (__all(p))
#define S2K_UNIFORM(x) (__builtin_amdgcn_readfirstlane(x))       /* a value known to be the same in every lane, as a scalar */
#else
#define S2K_WAVE_ANY(p) (p)
#define S2K_WAVE_ALL(p) (p)
#define S2K_UNIFORM(x) (x)
#endif

// Per-lane memory handed to ecmult_lane: the table slice in HBM and the digit stream in LDS.  The digit stream is what the
// main loop would otherwise carry in ~18 registers (two 160-bit digit shift registers and the 256-bit generator scalar) and
// spill around every point operation; in LDS it costs one ds_read per addition.  Word-major layout (word k of lane t at
// [k * S2K_DIG_STRIDE + t]) keeps the accesses bank-conflict free.
//   words 0..8 : 4-bit digit of addition a (2 <= a < 66) in nibble a & 7 of word a >> 3
//   words 9..17: ng recoded for the signed fixed-base windows (gtab_recode: 9 words); window g is bits [D g, D g + D)
#define S2K_DIG_WORDS 18
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) u32* s2k_lds_ptr;
#define S2K_DIG_STRIDE 256                  /* every kernel that calls ecmult_lane runs 256-lane workgroups */
#define S2K_LANE_DIG(shared_array) ((s2k_lds_ptr)(shared_array) + threadIdx.x)
#else
typedef u32* s2k_lds_ptr;
#define S2K_DIG_STRIDE 1
#define S2K_LANE_DIG(array) (array)
#endif
struct lane_mem { u32* ptab; s2k_lds_ptr dig; };

// 160-bit shift register holding a 129-bit magnitude: the next 4-bit digit window is always the top nibble
struct digit_reg { u32 w[5]; };
S2K_HD void digit_reg_init(digit_reg& r, const u32 k[5]) {          // k' << 31 : window of digit 31 (bits 125..128) at the top
    r.w[4] = (k[4] << 31) | (k[3] >> 1);
    r.w[3] = (k[3] << 31) | (k[2] >> 1);
    r.w[2] = (k[2] << 31) | (k[1] >> 1);
    r.w[1] = (k[1] << 31) | (k[0] >> 1);
    r.w[0] = (k[0] << 31);
}
S2K_HD u32 digit_reg_pop(digit_reg& r) {
    const u32 v = r.w[4] >> 28;
#pragma unroll
    for (int i = 4; i > 0; i--) r.w[i] = (r.w[i] << 4) | (r.w[i - 1] >> 28);
    r.w[0] <<= 4;
    return v;
}

#define S2K_ADDS_P 66            // 33 digit positions x 2 halves
#define S2K_ADD_G0 S2K_ADDS_P
#define S2K_ADDS_MAX (S2K_ADD_G0 + S2K_GTAB_MAX_WINDOWS)

// R = na*A + ng*G for this lane.  A is Jacobian (A.inf allowed), ng may be absent (has_ng = 0).
// gtab: generator table; ptab: this lane's private S2K_PTAB_WORDS-word slice of scratch memory.
// R is returned on the real curve, magnitudes (<=5,<=3,1).
S2K_HD void ecmult_lane(gej& R, const gej& A, const scalar& na, const scalar& ng, int has_ng, const u32* gtab, const lane_mem& lm) {
    u32* const ptab = lm.ptab; const s2k_lds_ptr dig = lm.dig;
    S2K_PROF_DECL;
    const int p_active = (!A.inf) & (!sc_is_zero(na));
    const int g_active = has_ng & (!sc_is_zero(ng));
    int hneg0, hneg1;
    fe ziso;
    {
        half_scalar h0, h1;
        sc_split_lambda_odd(h0, h1, na);                         // both magnitudes odd (< 2^129): no recoding correction
        hneg0 = h0.neg; hneg1 = h1.neg;
        {
            digit_reg dr0, dr1; digit_reg_init(dr0, h0.w); digit_reg_init(dr1, h1.w);
            u32 dw[9];
#pragma unroll
            for (int i = 0; i < 9; i++) dw[i] = 0;
#pragma unroll
            for (int i = 2; i < S2K_ADDS_P; i++) { const u32 v = (i & 1) ? digit_reg_pop(dr1) : digit_reg_pop(dr0); dw[i >> 3] |= v << ((i & 7) * 4); }
#pragma unroll
            for (int i = 0; i < 9; i++) dig[i * S2K_DIG_STRIDE] = dw[i];
        }
        S2K_PROF_MARK(1);
        if (S2K_WAVE_ANY(p_active)) {
            ptab_build(ziso, ptab, A);
#pragma unroll
            for (int i = 0; i < 9; i++) ptab[S2K_PTAB_ZISO + i] = ziso.n[i];
        }
    }
    const gtab_geom GG = gtab_geometry(gtab);
    {
        u32 ngr[S2K_GTAB_SWORDS]; gtab_recode(ngr, ng.d, gtab);
#pragma unroll
        for (int i = 0; i < S2K_GTAB_SWORDS; i++) dig[(9 + i) * S2K_DIG_STRIDE] = ngr[i];
    }

    S2K_PROF_MARK(2);
    // per-lane micro-program: additions a = 0..a_end-1, with 4 doublings in front of every even a in [2, 66)
    int a = p_active ? 0 : S2K_ADD_G0;
    int zfixed = !p_active;
    int dbl_left = 0, pending = 0;
    const int a_end = g_active ? S2K_ADD_G0 + (int)GG.W : S2K_ADD_G0;
    // Operand pipeline.  Every operand -- an odd multiple from this lane's table or a generator-table entry -- is one aligned
    // 64-byte record of canonical words (x, y).  `op_locate` turns an addition index into (record address, use it?, negate y?);
    // the record of addition a+1 is *requested* (16 words into `raw`, no use of the data) before the arithmetic of addition a
    // and *decoded* into limbs after it, so the gather's latency lies under ~1800 instructions instead of in front of them.
    // A lane that does not advance in a trip simply requests the same record again: no per-lane select ever waits on the data.
    auto op_locate = [&](const u32*& addr, int& valid, int& neg, int& lam, int idx) {
        addr = ptab; valid = 0; neg = 0; lam = 0;
        if (idx < S2K_ADD_G0) {
            const int h = idx & 1;
            const int hn = h ? hneg1 : hneg0;
            u32 v = 8u; int flip = hn; valid = 1;                                                   // top digit (a = 0, 1) is always +1
            if (idx >= 2 && idx < S2K_ADDS_P) v = (dig[(idx >> 3) * S2K_DIG_STRIDE] >> ((idx & 7) * 4)) & 15u;
            neg = (v < 8u) ^ flip;
            const u32 e = (v < 8u) ? (7u - v) : (v - 8u);
            addr = ptab + e * S2K_PTAB_ENTRY_WORDS + ((h && S2K_PTAB_TWINS) ? 16 : 0); lam = h && !S2K_PTAB_TWINS;
        } else if (idx < a_end) {
            const int g = idx - S2K_ADD_G0, w = (int)(((u32)g * GG.D) >> 5);
            valid = gtab_locate(addr, neg, gtab, GG, g, dig[(9 + w) * S2K_DIG_STRIDE], w + 1 < S2K_GTAB_SWORDS ? dig[(10 + w) * S2K_DIG_STRIDE] : 0u);
        }
    };
    auto op_decode = [&](ge& o, const u32 raw[16], int neg, int lam) {
        fe y, yn;
        fe_from_words(o.x, raw); fe_from_words(y, raw + 8);
        if (S2K_WAVE_ANY(lam)) {                                    // twin-less tables: the lambda half's x is beta * x
            fe beta, bx; fe_set_beta(beta); fe_mul(bx, o.x, beta);
            fe_select(o.x, bx, o.x, lam);
        }
        fe_neg(yn, y, 1);
        fe_select(o.y, yn, y, neg);
    };
    const u32* nxt_addr; int nxt_valid, nxt_neg, nxt_lam;
    u32 raw[16];
    ge cur; int cur_valid;
    op_locate(nxt_addr, nxt_valid, nxt_neg, nxt_lam, a);
#pragma unroll
    for (int k = 0; k < 16; k++) raw[k] = nxt_addr[k];
    op_decode(cur, raw, nxt_neg, nxt_lam); cur_valid = nxt_valid;
    gej_set_infinity(R);
    if (p_active) {                                 // addition 0 would add the top digit of half 0 to infinity: just take it
        gej_set_ge(R, cur);
        a = 1;
        op_locate(nxt_addr, nxt_valid, nxt_neg, nxt_lam, a);
#pragma unroll
        for (int k = 0; k < 16; k++) raw[k] = nxt_addr[k];
        op_decode(cur, raw, nxt_neg, nxt_lam); cur_valid = nxt_valid;
    }
    op_locate(nxt_addr, nxt_valid, nxt_neg, nxt_lam, a + 1);
    // Lock-step fast path.  When every lane of the wavefront multiplies a live point (and the lanes agree on whether there is a
    // generator part) they all sit at the same place of the same micro-program, so the schedule is a plain loop: no per-lane
    // program counter, no select after every point operation, lean formulas (group.h).  A lane that meets an operand with its
    // own x coordinate (P + P, P - P: adversarial inputs only) makes the wavefront leave the fast path *before* that addition
    // is committed; the general loop below picks up from exactly that state.
#ifndef S2K_NO_FAST_PATH
    if (S2K_WAVE_ALL(p_active) && (S2K_WAVE_ALL(g_active) || !S2K_WAVE_ANY(g_active))) {
        int au = 1;                                              // uniform copy of `a`
        while (au < a_end) {
            if (au >= 2 && au < S2K_ADDS_P && !(au & 1)) {
#pragma unroll 1
                for (int k = 0; k < 4; k++) gej_double_lean(R, R);
                S2K_PROF_MARK(6);
            }
            if (au == S2K_ADD_G0) { fe zi; ptab_load_ziso(zi, ptab); fe_mul(R.z, R.z, zi); zfixed = 1; }      // back to the real curve
#pragma unroll
            for (int k = 0; k < 16; k++) raw[k] = nxt_addr[k];                  // request the next record before the arithmetic
            gej t; const int same_x = gej_add_ge_lean(t, R, cur);
            if (S2K_WAVE_ANY(same_x & cur_valid)) break;                        // -> general loop, nothing committed
            if (au < S2K_ADDS_P) R = t;                                         // regular digits are never zero
            else if (cur_valid) R = t;                                          // generator windows: a zero window adds nothing (per lane)
            au++;
            op_decode(cur, raw, nxt_neg, nxt_lam); cur_valid = nxt_valid;
            op_locate(nxt_addr, nxt_valid, nxt_neg, nxt_lam, au + 1);
        }
        if (au == S2K_ADD_G0 && !zfixed) { fe zi; ptab_load_ziso(zi, ptab); fe_mul(R.z, R.z, zi); zfixed = 1; }      // no generator part
        a = au;
    }
#endif
    int done = !(a < a_end) & zfixed;
    while (S2K_WAVE_ANY(!done)) {
        int do_dbl = 0, do_add = 0;
        if (!done) {
            if (pending) { do_dbl = 1; pending = 0; }
            else if (dbl_left > 0) { do_dbl = !R.inf; dbl_left--; }
            else do_add = 1;
        }
        if (S2K_WAVE_ANY(do_dbl)) {
            gej t; gej_double(t, R);
            if (do_dbl) R = t;
        }
        if (S2K_WAVE_ANY(do_add)) {
#pragma unroll
            for (int k = 0; k < 16; k++) raw[k] = nxt_addr[k];                  // request the next record before the arithmetic
            gej t; const int f = gej_add_ge(t, R, cur);
            ge nx; op_decode(nx, raw, nxt_neg, nxt_lam);
            if (do_add) {
                if (cur_valid) { R = t; pending = (f == GEJ_ADD_NEEDS_DOUBLE); }
                a++;
                cur = nx; cur_valid = nxt_valid;
                dbl_left = (a >= 2 && a < S2K_ADDS_P && !(a & 1)) ? 4 : 0;
                op_locate(nxt_addr, nxt_valid, nxt_neg, nxt_lam, a + 1);
            }
        }
        // leaving the isomorphic curve: after the last digit, before the generator additions
        const int fix = (!done) & (!zfixed) & (a == S2K_ADD_G0) & (!pending);
        if (S2K_WAVE_ANY(fix)) {
            fe zi;
#pragma unroll
            for (int i = 0; i < 9; i++) zi.n[i] = ptab[S2K_PTAB_ZISO + i];
            fe z; fe_mul(z, R.z, zi);
            if (fix) { R.z = z; zfixed = 1; }
        }
        if ((a >= a_end) & (!pending) & zfixed) done = 1;
    }
    S2K_PROF_MARK(3);
}


// ---- the same multiplication with the variable part cut in two: R = na*A + ng*G given also T = 2^64 * A -------------------------
// When 2^64*A is known (the rangeproof rings keep it next to every ring key: one 64-doubling chain per ring, then the same
// "+ base" step as the key itself, rangeproof.h), each odd GLV half k = ka + 2^64 kb splits into two 65-bit odd pieces and
//     na*A = (+-k1a) A + (+-k2a) lambda A + (+-k1b) T + (+-k2b) lambda T
// needs 64 doublings instead of 128, for one more table and two more additions: four digit streams of 17 signed odd 4-bit
// digits, 4 doublings + 4 additions per digit position.  Both tables are rescaled to ONE common Z (ptab_rescale), so the
// accumulator sits on one isomorphic curve as before.  Only the lock-step form exists: the function returns 0 without having
// produced anything when the wavefront is not uniform or a lane meets an operand with its own x coordinate, and the caller then
// runs ecmult_lane.  Digit stream in LDS: words 0..7 = nibble (pos * 4 + stream), words 8..16 = ng recoded (gtab_recode).
#define S2K_SPLIT_ADDS_P 68          // 4 streams x 17 digits
struct piece65 { u32 w[3]; int neg; };
S2K_HD void sc_split_pieces(piece65 out[4], const half_scalar& h0, const half_scalar& h1) {
    // out[0] = low piece of k1, out[1] = low piece of k2, out[2] = high piece of k1, out[3] = high piece of k2 (stream order)
    for (int hf = 0; hf < 2; hf++) {
        const half_scalar& h = hf ? h1 : h0;
        u32 lo[3] = {h.w[0], h.w[1], 0u};
        u32 hi[3] = {h.w[2], h.w[3], h.w[4]};                        // k >> 64 (< 2^65)
        int lo_neg = h.neg;
        if (!(hi[0] & 1u)) {                                         // make the high piece odd: hi += 1, lo -= 2^64 (lo becomes negative)
            u32 c = 1u;
            for (int i = 0; i < 3; i++) { const u32 t = hi[i] + c; c = (t < c); hi[i] = t; }
            // |lo - 2^64| = 2^64 - lo  (lo odd, so nonzero)
            const u64 l = (u64)lo[0] | ((u64)lo[1] << 32);
            const u64 m = 0ull - l;
            lo[0] = (u32)m; lo[1] = (u32)(m >> 32); lo[2] = 0u;
            lo_neg = !h.neg;
        }
        for (int i = 0; i < 3; i++) { out[hf].w[i] = lo[i]; out[2 + hf].w[i] = hi[i]; }
        out[hf].neg = lo_neg; out[2 + hf].neg = h.neg;
    }
}
S2K_HD int ecmult_lane_split(gej& R, const gej& A, const gej& T, const scalar& na, const scalar& ng, int has_ng, const u32* gtab, const lane_mem& lm) {
    u32* const ptab = lm.ptab; const s2k_lds_ptr dig = lm.dig;
    const int p_active = (!A.inf) & (!T.inf) & (!sc_is_zero(na));
    const int g_active = has_ng & (!sc_is_zero(ng));
    if (!(S2K_WAVE_ALL(p_active) && (S2K_WAVE_ALL(g_active) || !S2K_WAVE_ANY(g_active)))) return 0;
    u32 sneg = 0;                                                    // bit st: stream st is negative
    S2K_PROF_DECL;
    {
        half_scalar h0, h1; sc_split_lambda_odd(h0, h1, na);
        piece65 pc[4]; sc_split_pieces(pc, h0, h1);
        u32 dw[8];
#pragma unroll
        for (int i = 0; i < 8; i++) dw[i] = 0;
#pragma unroll
        for (int st = 0; st < 4; st++) {
            sneg |= (u32)pc[st].neg << st;
#pragma unroll
            for (int pos = 0; pos < 16; pos++) {                     // pos 0 = digit 15 (most significant after the fixed top digit)
                const int i = 15 - pos, bit = 4 * i + 1, word = bit >> 5, sh = bit & 31;
                const u64 pair = (u64)pc[st].w[word] | ((u64)(word + 1 < 3 ? pc[st].w[word + 1] : 0u) << 32);
                const u32 v = (u32)(pair >> sh) & 15u;
                const int nib = pos * 4 + st;
                dw[nib >> 3] |= v << ((nib & 7) * 4);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) dig[i * S2K_DIG_STRIDE] = dw[i];
        {
            u32 ngr[S2K_GTAB_SWORDS]; gtab_recode(ngr, ng.d, gtab);
#pragma unroll
            for (int i = 0; i < S2K_GTAB_SWORDS; i++) dig[(8 + i) * S2K_DIG_STRIDE] = ngr[i];
        }
        fe za, zt, ziso;
        S2K_PROF_MARK(1);
        ptab_build_raw(za, ptab, A);
        ptab_build_raw(zt, ptab + S2K_PTAB_TABLE_WORDS, T);
        S2K_PROF_MARK(8);
        ptab_rescale(ptab, &zt);
        ptab_rescale(ptab + S2K_PTAB_TABLE_WORDS, &za);
        S2K_PROF_MARK(9);
        fe_mul(ziso, za, zt);
#pragma unroll
        for (int i = 0; i < 9; i++) ptab[S2K_PTAB_ZISO + i] = ziso.n[i];
    }
    const gtab_geom GG = gtab_geometry(gtab);
    const int a_g0 = S2K_SPLIT_ADDS_P;
    const int a_end = g_active ? a_g0 + (int)GG.W : a_g0;
    auto op_locate = [&](const u32*& addr, int& valid, int& neg, int& lam, int idx) {
        addr = ptab; valid = 0; neg = 0; lam = 0;
        if (idx < a_g0) {
            const int st = idx & 3;
            u32 v = 8u;                                                                     // the fixed top digit +1
            if (idx >= 4) { const int nib = idx - 4; v = (dig[(nib >> 3) * S2K_DIG_STRIDE] >> ((nib & 7) * 4)) & 15u; }
            valid = 1;
            neg = (v < 8u) ^ (int)((sneg >> st) & 1u);
            const u32 e = (v < 8u) ? (7u - v) : (v - 8u);
            addr = ptab + (st >> 1) * S2K_PTAB_TABLE_WORDS + e * S2K_PTAB_ENTRY_WORDS + (((st & 1) && S2K_PTAB_TWINS) ? 16 : 0); lam = (st & 1) && !S2K_PTAB_TWINS;
        } else if (idx < a_end) {
            const int g = idx - a_g0, w = (int)(((u32)g * GG.D) >> 5);
            valid = gtab_locate(addr, neg, gtab, GG, g, dig[(8 + w) * S2K_DIG_STRIDE], w + 1 < S2K_GTAB_SWORDS ? dig[(9 + w) * S2K_DIG_STRIDE] : 0u);
        }
    };
    auto op_decode = [&](ge& o, const u32 raw[16], int neg, int lam) {
        fe y, yn;
        fe_from_words(o.x, raw); fe_from_words(y, raw + 8);
        if (lam) { fe beta; fe_set_beta(beta); fe_mul(o.x, o.x, beta); }          // `lam` is a compile-time constant at every call site
        fe_neg(yn, y, 1);
        fe_select(o.y, yn, y, neg);
    };
    const u32* nxt_addr; int nxt_valid, nxt_neg, nxt_lam;
    u32 raw[16];
    ge cur; int cur_valid;
    op_locate(nxt_addr, nxt_valid, nxt_neg, nxt_lam, 0);
#pragma unroll
    for (int k = 0; k < 16; k++) raw[k] = nxt_addr[k];
    op_decode(cur, raw, nxt_neg, 0);
    op_locate(nxt_addr, nxt_valid, nxt_neg, nxt_lam, 1);
    // Variable part: additions 0..67 as 34 (plain stream, lambda stream) pairs -- the trip holds both additions, so which operand
    // takes the beta product is static and the accumulator can alternate between two register sets instead of being copied back at
    // every loop edge.  Addition 0 adds to infinity: it just takes its operand.  In place: a lane that meets its own x coordinate
    // makes the caller start over with ecmult_lane, so the accumulator of before the addition need not survive it.
    int au = 0;
    while (au < a_g0) {
        if (au >= 4 && !(au & 3)) {
#pragma unroll 1
            for (int k = 0; k < 4; k++) gej_double_lean(R, R);
            S2K_PROF_MARK(6);
        }
#pragma unroll
        for (int k = 0; k < 16; k++) raw[k] = nxt_addr[k];                  // request the next record before the arithmetic
        if (au == 0) gej_set_ge(R, cur);
        else { const int same_x = gej_add_ge_lean(R, R, cur); if (S2K_WAVE_ANY(same_x)) return 0; }
        op_decode(cur, raw, nxt_neg, !S2K_PTAB_TWINS);                      // operand of the lambda stream
        op_locate(nxt_addr, nxt_valid, nxt_neg, nxt_lam, au + 2);           // (au = 66: the first generator window, if any)
#pragma unroll
        for (int k = 0; k < 16; k++) raw[k] = nxt_addr[k];
        { const int same_x = gej_add_ge_lean(R, R, cur); if (S2K_WAVE_ANY(same_x)) return 0; }
        au += 2;
        op_decode(cur, raw, nxt_neg, 0); cur_valid = nxt_valid;
        op_locate(nxt_addr, nxt_valid, nxt_neg, nxt_lam, au + 1);
    }
    { fe zi; ptab_load_ziso(zi, ptab); fe_mul(R.z, R.z, zi); }             // back to the real curve
    // generator part: a zero window adds nothing (per lane), so these additions are committed by select
#if S2K_XYZZ_TABLE_PART
    gez acc4; acc4.inf = 0; acc4.x = R.x; acc4.y = R.y;
    fe_sqr(acc4.zz, R.z); fe_mul(acc4.zzz, acc4.zz, R.z);
#endif
    while (au < a_end) {
#if S2K_NT_GTAB_SPLIT && defined(__HIP_DEVICE_COMPILE__)
        {   // (the fixed-base operands are touched once: a non-temporal request keeps them from displacing the per-lane tables)
            typedef unsigned int s2k_u32x4 __attribute__((ext_vector_type(4)));
            const s2k_u32x4* q = (const s2k_u32x4*)nxt_addr;
#pragma unroll
            for (int k = 0; k < 4; k++) { const s2k_u32x4 v = __builtin_nontemporal_load(q + k); raw[4 * k] = v.x; raw[4 * k + 1] = v.y; raw[4 * k + 2] = v.z; raw[4 * k + 3] = v.w; }
        }
#else
#pragma unroll
        for (int k = 0; k < 16; k++) raw[k] = nxt_addr[k];
#endif
#if S2K_XYZZ_TABLE_PART
        {
            gez t = acc4; gez_add_ge_lean(t, cur);
            if (S2K_WAVE_ALL(cur_valid)) acc4 = t;               // (a zero digit -- one window value in 2^D -- is the only reason for the selects)
            else { fe_cmov(acc4.x, t.x, cur_valid); fe_cmov(acc4.y, t.y, cur_valid); fe_cmov(acc4.zz, t.zz, cur_valid); fe_cmov(acc4.zzz, t.zzz, cur_valid); }
        }
#else
        gej t; const int same_x = gej_add_ge_lean(t, R, cur);
        if (S2K_WAVE_ANY(same_x & cur_valid)) return 0;
        if (cur_valid) R = t;
#endif
        au++;
        op_decode(cur, raw, nxt_neg, nxt_lam); cur_valid = nxt_valid;
        op_locate(nxt_addr, nxt_valid, nxt_neg, nxt_lam, au + 1);
    }
#if S2K_XYZZ_TABLE_PART
    if (S2K_WAVE_ANY(fe_normalizes_to_zero(acc4.zz))) return 0;
    gej_set_gez(R, acc4);
#endif
    S2K_PROF_MARK(10);
#ifdef S2K_ON_SPLIT_DONE
    S2K_ON_SPLIT_DONE();                                                                      // host test build: count completed split runs
#endif
    return 1;
}


// ---- the ring form: several multiplications by the SAME point  R_j = e_j*C + s_j*G + f_j*H ---------------------------------------
// A Borromean ring verifies four public keys P_j = C + j*B with B = -(4^i 10^exp)*H a known multiple of the proof's generator H
// (secp256k1_rangeproof_pub_expand, rangeproof_impl.h:19-51), so   e_j*P_j = e_j*C + f_j*H ,  f_j = -(j 4^i 10^exp) e_j mod n :
// the variable point is the same for all four steps of the ring, and the part that changes goes through a fixed-base table of H laid
// out exactly like the one of G (gtable.h; the engine keeps a small cache of them keyed by the generator's 64 bytes).  What that buys:
//   * the two odd-multiples tables (of C and of T = 2^64*C) and the 64-doubling chain are built ONCE per ring instead of once per step,
//     so they can be twice as large: signed odd 5-bit digits, 16 entries per table, 13 additions per stream instead of 17.
//     (13, not 14: 13 digits d_i = 2 b_i - 31 represent every odd k with |k| < 2^65 -- sum d_i 32^i = 2B - (2^65 - 1), so b is read off
//     B = (k - 1)/2 + 2^64: bit t of B is bit t + 1 of k and bit 64 is set, i.e. the top digit is 16 | (k >> 61) -- and the four 65-bit
//     pieces of sc_split_pieces are odd and below 2^65.  ecmult_lane_split's 4-bit digits need a fixed 17th digit because 16 of them
//     only reach 2^64.)  So a step is 12 x 5 doublings and 52 additions, the first of which just takes its operand;
//   * no "key <- key + B", "T <- T + 2^64*B" updates between the steps;
//   * + W (the table's number of windows) additions from H's table on the steps with j > 0.
// Per ring: 1 chain + 2 tables + 4 x (60 doublings + 52 additions) + 77 table additions, against 4 x (64 doublings + 68 + 11 additions
// + 2 tables + a chain quarter + 2 key updates) for ecmult_lane_split.
// Lane memory: `rtab`, S2K_RTAB_WORDS words of HBM: 32 finished 64-byte sectors back to back (2 KB: all the main loop touches) and the Z
// factor; the parked entries of the construction live in a per-wavefront, lane-interleaved area (S2K_RRAW_WAVE_WORDS).
// Digit stream in LDS (S2K_RING_DIG_WORDS words per lane): words 0..8 = 5-bit digit (pos * 4 + stream), six per word; 9..17 = s and 18..26 = f,
// both recoded for the signed fixed-base windows (gtab_recode).
// Only the lock-step form exists (every lane of the wavefront works: the caller gives idle lanes a dummy point and dummy scalars);
// ecmult_ring_step returns 0, having produced nothing, when a lane met an operand with its own x coordinate, and the caller then takes
// that step through ecmult_lane on P_j itself.
#define S2K_RING_W 5
#define S2K_RING_ENTRIES 16
#define S2K_RING_DIGITS 13                                      /* 13 signed odd 5-bit digits: every odd |k| < 2^65, no extra top digit (below) */
#define S2K_RING_ADDS_P (4 * S2K_RING_DIGITS)
#define S2K_RING_DIG_WORDS 27
#define S2K_RTAB_TABLE_WORDS (S2K_RING_ENTRIES * 16)
#define S2K_RTAB_ZISO (2 * S2K_RTAB_TABLE_WORDS)
#define S2K_RTAB_WORDS (S2K_RTAB_ZISO + 16)                     /* per lane: 32 finished sectors + the Z factor */
// parking area of the construction: per WAVEFRONT, word k of parked entry e of the 64 lanes at ((e * 27 + k) * 64 + lane)
#if defined(__HIP_DEVICE_COMPILE__)
#define S2K_RAW_WS 64
#else
#define S2K_RAW_WS 1
#endif
#define S2K_RAW_ES (27 * S2K_RAW_WS)
#define S2K_RRAW_WAVE_WORDS (2 * S2K_RING_ENTRIES * 27 * 64)

// C, T = 2^64*C finite, magnitudes <= (5,3,1).  rtab: this lane's S2K_RTAB_WORDS; raw: this lane's column of its wavefront's parking area
S2K_HD void ecmult_ring_tables(u32* rtab, u32* raw, const gej& C, const gej& T) {
    fe za, zt, ziso;
    u32* const raw_t = raw + S2K_RING_ENTRIES * S2K_RAW_ES;
    ptab_build_raw_n<S2K_RING_ENTRIES, S2K_RAW_WS, S2K_RAW_ES>(za, raw, C);
    ptab_build_raw_n<S2K_RING_ENTRIES, S2K_RAW_WS, S2K_RAW_ES>(zt, raw_t, T);
    ptab_rescale_n<S2K_RING_ENTRIES, S2K_RAW_WS, S2K_RAW_ES>(raw, rtab, 16, &zt);
    ptab_rescale_n<S2K_RING_ENTRIES, S2K_RAW_WS, S2K_RAW_ES>(raw_t, rtab + S2K_RTAB_TABLE_WORDS, 16, &za);
    fe_mul(ziso, za, zt);
#pragma unroll
    for (int i = 0; i < 9; i++) rtab[S2K_RTAB_ZISO + i] = ziso.n[i];
}
// e != 0; has_f uniform over the wavefront.  htab: this lane's generator's table (same layout as gtab).
S2K_HD int ecmult_ring_step(gej& R, const u32* rtab, const scalar& e, const scalar& s, const scalar& f, int has_f, const u32* gtab, const u32* htab,
                            const s2k_lds_ptr dig) {
    u32 sneg = 0;
    S2K_PROF_DECL;
    {
        half_scalar h0, h1; sc_split_lambda_odd(h0, h1, e);
        piece65 pc[4]; sc_split_pieces(pc, h0, h1);
        u32 dw[9];
#pragma unroll
        for (int i = 0; i < 9; i++) dw[i] = 0;
#pragma unroll
        for (int st = 0; st < 4; st++) {
            sneg |= (u32)pc[st].neg << st;
#pragma unroll
            for (int pos = 0; pos < S2K_RING_DIGITS; pos++) {            // pos 0 = most significant digit
                const int i = S2K_RING_DIGITS - 1 - pos, bit = S2K_RING_W * i + 1, word = bit >> 5, sh = bit & 31;
                const u64 pair = (u64)pc[st].w[word] | ((u64)(word + 1 < 3 ? pc[st].w[word + 1] : 0u) << 32);
                u32 v = (u32)(pair >> sh) & 31u;
                if (pos == 0) v |= 16u;                                  // bit 64 of B (the piece is below 2^65: bits 65, 66 are clear)
                const int nib = pos * 4 + st;
                dw[nib / 6] |= v << ((nib % 6) * 5);
            }
        }
#pragma unroll
        for (int i = 0; i < 9; i++) dig[i * S2K_DIG_STRIDE] = dw[i];
        u32 sr[S2K_GTAB_SWORDS], fr[S2K_GTAB_SWORDS]; gtab_recode(sr, s.d, gtab); gtab_recode(fr, f.d, has_f ? htab : gtab);
#pragma unroll
        for (int i = 0; i < S2K_GTAB_SWORDS; i++) { dig[(9 + i) * S2K_DIG_STRIDE] = sr[i]; dig[(18 + i) * S2K_DIG_STRIDE] = fr[i]; }
    }
    S2K_PROF_MARK(1);
    const gtab_geom GG = gtab_geometry(gtab), GH = gtab_geometry(has_f ? htab : gtab);      // (the generator's table may have another width than G's)
    const int a_g0 = S2K_RING_ADDS_P, a_h0 = a_g0 + (int)GG.W;
    const int a_end = has_f ? a_h0 + (int)GH.W : a_h0;
    auto op_locate = [&](const u32*& addr, int& valid, int& neg, int idx) {
        addr = rtab; valid = 0; neg = 0;
        if (idx < a_g0) {
            const int st = idx & 3;
            const u32 v = (dig[(idx / 6) * S2K_DIG_STRIDE] >> ((idx % 6) * 5)) & 31u;
            valid = 1;
            neg = (v < 16u) ^ (int)((sneg >> st) & 1u);
            const u32 en = (v < 16u) ? (15u - v) : (v - 16u);
            addr = rtab + (st >> 1) * S2K_RTAB_TABLE_WORDS + en * 16;
        } else if (idx < a_end) {
            const int second = idx >= a_h0;
            const int g = idx - (second ? a_h0 : a_g0), base = second ? 18 : 9, w = (int)(((u32)g * (second ? GH.D : GG.D)) >> 5);
            valid = gtab_locate(addr, neg, second ? htab : gtab, second ? GH : GG, g, dig[(base + w) * S2K_DIG_STRIDE], w + 1 < S2K_GTAB_SWORDS ? dig[(base + 1 + w) * S2K_DIG_STRIDE] : 0u);
        }
    };
    auto op_decode = [&](ge& o, const u32 raw[16], int neg, int lam) {
        fe y, yn;
        fe_from_words(o.x, raw); fe_from_words(y, raw + 8);
        if (lam) { fe beta; fe_set_beta(beta); fe_mul(o.x, o.x, beta); }          // `lam` is a compile-time constant at every call site
        fe_neg(yn, y, 1);
        fe_select(o.y, yn, y, neg);
    };
    const u32* nxt_addr; int nxt_valid, nxt_neg;
    u32 raw[16];
    ge cur; int cur_valid;
    op_locate(nxt_addr, nxt_valid, nxt_neg, 0);
#pragma unroll
    for (int k = 0; k < 16; k++) raw[k] = nxt_addr[k];
    op_decode(cur, raw, nxt_neg, 0);
    op_locate(nxt_addr, nxt_valid, nxt_neg, 1);
    // variable part: additions 0..51 as 26 (plain stream, lambda stream) pairs, 5 doublings in front of every group of four but the first
    int au = 0;
    while (au < a_g0) {
        if (au >= 4 && !(au & 3)) {
            S2K_PROF_MARK(7);
#pragma unroll 1
            for (int k = 0; k < S2K_RING_W; k++) gej_double_lean(R, R);
            S2K_PROF_MARK(6);
        }
#pragma unroll
        for (int k = 0; k < 16; k++) raw[k] = nxt_addr[k];                  // request the next record before the arithmetic
        if (au == 0) gej_set_ge(R, cur);
        else { const int same_x = gej_add_ge_lean(R, R, cur); if (S2K_WAVE_ANY(same_x)) return 0; }
        op_decode(cur, raw, nxt_neg, 1);                                    // operand of the lambda stream
        op_locate(nxt_addr, nxt_valid, nxt_neg, au + 2);
#pragma unroll
        for (int k = 0; k < 16; k++) raw[k] = nxt_addr[k];
        { const int same_x = gej_add_ge_lean(R, R, cur); if (S2K_WAVE_ANY(same_x)) return 0; }
        au += 2;
        op_decode(cur, raw, nxt_neg, 0); cur_valid = nxt_valid;
        op_locate(nxt_addr, nxt_valid, nxt_neg, au + 1);
    }
    S2K_PROF_MARK(7);
    {   // back to the real curve
        fe zi;
#pragma unroll
        for (int i = 0; i < 9; i++) zi.n[i] = rtab[S2K_RTAB_ZISO + i];
        fe_mul(R.z, R.z, zi);
    }
    // table part (G, then H): a zero window adds nothing (per lane), so these additions are committed by select
#if S2K_XYZZ_TABLE_PART
    gez acc4; acc4.inf = 0; acc4.x = R.x; acc4.y = R.y;
    fe_sqr(acc4.zz, R.z); fe_mul(acc4.zzz, acc4.zz, R.z);
#endif
    while (au < a_end) {
#if S2K_NT_GTAB && defined(__HIP_DEVICE_COMPILE__)
        {
            typedef unsigned int s2k_u32x4 __attribute__((ext_vector_type(4)));
            const s2k_u32x4* q = (const s2k_u32x4*)nxt_addr;
#pragma unroll
            for (int k = 0; k < 4; k++) { const s2k_u32x4 v = __builtin_nontemporal_load(q + k); raw[4 * k] = v.x; raw[4 * k + 1] = v.y; raw[4 * k + 2] = v.z; raw[4 * k + 3] = v.w; }
        }
#else
#pragma unroll
        for (int k = 0; k < 16; k++) raw[k] = nxt_addr[k];
#endif
#if S2K_XYZZ_TABLE_PART
        {
            gez t = acc4; gez_add_ge_lean(t, cur);
            if (S2K_WAVE_ALL(cur_valid)) acc4 = t;               // (a zero digit -- one window value in 2^D -- is the only reason for the selects)
            else { fe_cmov(acc4.x, t.x, cur_valid); fe_cmov(acc4.y, t.y, cur_valid); fe_cmov(acc4.zz, t.zz, cur_valid); fe_cmov(acc4.zzz, t.zzz, cur_valid); }
        }
#else
        gej t; const int same_x = gej_add_ge_lean(t, R, cur);
        if (S2K_WAVE_ANY(same_x & cur_valid)) return 0;
        if (cur_valid) R = t;
#endif
        au++;
        op_decode(cur, raw, nxt_neg, 0); cur_valid = nxt_valid;
        op_locate(nxt_addr, nxt_valid, nxt_neg, au + 1);
    }
#if S2K_XYZZ_TABLE_PART
    if (S2K_WAVE_ANY(fe_normalizes_to_zero(acc4.zz))) return 0;      // some committed addition met an operand with the accumulator's own x
    gej_set_gez(R, acc4);
#endif
    S2K_PROF_MARK(10);
#ifdef S2K_ON_RING_STEP_DONE
    S2K_ON_RING_STEP_DONE();
#endif
    return 1;
}
