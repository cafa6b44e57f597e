// Which fused tile programs exist: X(tile rows, tile cols, tap lengths...).
// Level 1 keys are (len h0o, len h1o) forward and (len g0o, len g1o) inverse of the
// shipped biort sets; level >= 2 keys are the q-shift length.  Anything else goes
// through the generic filters (still on the GPU).
#pragma once
#define DT_FWD1_TABLE(X) \
    X(32, 64, 5, 7)      /* near_sym_a */ \
    X(32, 64, 9, 7)      /* antonini   */ \
    X(32, 64, 5, 3)      /* legall     */ \
    X(32, 64, 13, 19)    /* near_sym_b */
#define DT_INV1_TABLE(X) \
    X(32, 32, 7, 5) \
    X(32, 32, 7, 9) \
    X(32, 32, 3, 5) \
    X(32, 32, 19, 13)
#define DT_FWD2_TABLE(X) \
    X(32, 32, 10)        /* qshift_a, qshift_06 */ \
    X(32, 32, 14)        /* qshift_b */ \
    X(32, 32, 16)        /* qshift_c */ \
    X(32, 32, 18)        /* qshift_d */ \
    X(16, 16, 32)        /* qshift_32 */
#define DT_INV2_TABLE(X) \
    X(32, 32, 10) \
    X(32, 32, 14) \
    X(32, 32, 16) \
    X(32, 32, 18) \
    X(16, 16, 32)
