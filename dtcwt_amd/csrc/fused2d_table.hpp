// Which fused tile programs exist: X(tile rows, tile cols, tap lengths...).
// Level 1 keys are (len h0o, len h1o) forward and (len g0o, len g1o) inverse of the
// shipped biort sets; level >= 2 keys are the q-shift length.  Anything else goes
// through the generic filters (still on the GPU).
#pragma once
/* level-1 forward: X(tile rows, tile cols, rows per column-pass strip, len h0o, len h1o) */
#define DT_FWD1_TABLE(X) \
    X(32, 64, 8, 5, 7)      /* near_sym_a */ \
    X(32, 64, 8, 9, 7)      /* antonini   */ \
    X(32, 64, 8, 5, 3)      /* legall     */ \
    X(32, 64, 8, 13, 19)    /* near_sym_b */
/* level-1 inverse: X(tile rows, tile cols, rows per column-pass strip, len g0o, len g1o);
 * tile cols + 2*even-halo = 128 columns: (strips x column pairs) = one column-pass task per
 * thread, parity uniform per wavefront */
#define DT_INV1_TABLE(X) \
    X(16, 120, 8, 7, 5)      /* near_sym_a */ \
    X(16, 120, 8, 7, 9)      /* antonini   */ \
    X(16, 124, 8, 3, 5)      /* legall     */ \
    X(16, 108, 8, 19, 13)    /* near_sym_b */
/* level >= 2 forward: X(tile rows, tile cols, (A,B) pairs per column-pass strip, q-shift length);
 * tile cols chosen so that the input window 2*TC + 2*M - 4 is 128 columns (2 strips x 128 =
 * one task per thread in the column pass) */
#define DT_FWD2_TABLE(X) \
    X(16, 56, 4, 10)        /* qshift_a, qshift_06 */ \
    X(16, 52, 4, 14)        /* qshift_b */ \
    X(16, 50, 4, 16)        /* qshift_c */ \
    X(16, 48, 4, 18)        /* qshift_d */ \
    X(16, 34, 2, 32)        /* qshift_32 */
/* level >= 2 inverse: X(tile rows, tile cols (INPUT samples), j's per column-pass strip,
 * q-shift length); tile cols + window - 2 = 64 columns */
/* measured and dropped (profiles/r02/ab_inv2_tiles.txt): 32 x 56 and 16 x 120 tiles (record window 1.43 x /
 * 1.6 x the compulsory records instead of 1.71 x): level 2 unchanged at 32.5 us, levels 3-4 slower */
#define DT_INV2_TABLE(X) \
    X(16, 56, 2, 10) \
    X(16, 52, 2, 14) \
    X(16, 48, 2, 16) \
    X(16, 48, 2, 18) \
    X(16, 32, 2, 32)

/* Band-pass variants (6-vector biort / 12-vector q-shift sets: a third filter for the diagonal
 * subbands, dtcwt/numpy/transform2d.py:116-129, :145-155, :250-271, :283-291).
 * level 1: X(tile rows, tile cols, strip, len lo, len hi, len band-pass) */
#define DT_FWD1_BP_TABLE(X) \
    X(32, 64, 8, 13, 19, 19)     /* near_sym_b_bp: h0o h1o h2o */
#define DT_INV1_BP_TABLE(X) \
    X(16, 108, 8, 19, 13, 19)    /* near_sym_b_bp: g0o g1o g2o */
#define DT_FWD2_BP_TABLE(X) \
    X(16, 52, 4, 14)             /* qshift_b_bp */
#define DT_INV2_BP_TABLE(X) \
    X(16, 52, 2, 14)

/* Smaller tiles for coarse levels (more workgroups).  Measured on MI355X (4096^2, levels 3-4): no gain for the forward --
 * those launches sit at a ~3-8 us floor either way -- so the forward has none (its 8-row builds and DTCWT_HIP_SMALL_TILES
 * went in round 6).  The inverse does gain at the very coarsest levels, with 8 x 64 tiles: below this many 16 x 56 tiles
 * (two per CU; measured: a 512^2 lowpass, 320 tiles: 4.7 -> 3.7 us; a 1024^2 one, 1216 tiles: 10.4 -> 11.7 us) */
#ifndef DT_INV2_SMALL_BELOW
#define DT_INV2_SMALL_BELOW 512
#endif
#define DT_INV2_SMALL_TABLE(X) \
    X(8, 64, 2, 10) \
    X(8, 20, 2, 14) \
    X(8, 16, 2, 16) \
    X(8, 16, 2, 18) \
    X(16, 32, 2, 32)
