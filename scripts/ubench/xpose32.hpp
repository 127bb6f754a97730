// EXPERIMENT (not part of the product): in the fused kernel this replaced the LDS transpose between
// the two radix-32 stages; results were identical but the unit took 21.8k instead of 18.6k cycles,
// because a lone wave issues one VALU instruction per ~4.7 cycles and the 640 extra selects cost more
// than the LDS round trip they remove.
// xpose32.hpp -- transpose a 32 x 32 matrix held as 32 registers x 32 lanes (each half of a
// wave64 separately) without touching LDS: five rounds, round b swaps register-index bit b
// with lane bit b.  For a register pair (A = r[i], B = r[i | m]) and the lane pair (l, l ^ m)
// that is   A' = lanebit ? B[l ^ m] : A,   B' = lanebit ? B : A[l ^ m],
// one v_cndmask_b32_dpp each (bits 0..3: quad_perm / row_shr,row_shl / row_ror) and one
// v_permlane16_swap_b32 per pair for bit 4.
#pragma once
#include <hip/hip_runtime.h>

namespace af {
namespace feat {

// Four register pairs per asm block: VCC = lanes with the bit clear for the A' selects,
// inverted for the B' selects.  The two s_mov also are the two wait states a DPP read needs
// after a VALU write of its source.
#define AF_XCHG4(CTRL_HI, CTRL_LO, a0, b0, a1, b1, a2, b2, a3, b3, mask_lo)                           \
    do {                                                                                              \
        float _a0, _b0, _a1, _b1, _a2, _b2, _a3, _b3;                                                 \
        asm volatile(                                                                                 \
            "s_mov_b32 vcc_lo, %16\n\t"                                                               \
            "s_mov_b32 vcc_hi, %16\n\t"                                                                           \
            "v_cndmask_b32_dpp %0, %9, %8, vcc " CTRL_HI " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"  \
            "v_cndmask_b32_dpp %2, %11, %10, vcc " CTRL_HI " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
            "v_cndmask_b32_dpp %4, %13, %12, vcc " CTRL_HI " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
            "v_cndmask_b32_dpp %6, %15, %14, vcc " CTRL_HI " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
            "s_not_b64 vcc, vcc\n\t"                                                                  \
            "v_cndmask_b32_dpp %1, %8, %9, vcc " CTRL_LO " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"  \
            "v_cndmask_b32_dpp %3, %10, %11, vcc " CTRL_LO " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
            "v_cndmask_b32_dpp %5, %12, %13, vcc " CTRL_LO " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
            "v_cndmask_b32_dpp %7, %14, %15, vcc " CTRL_LO " row_mask:0xf bank_mask:0xf bound_ctrl:1"    \
            : "=&v"(_a0), "=&v"(_b0), "=&v"(_a1), "=&v"(_b1), "=&v"(_a2), "=&v"(_b2), "=&v"(_a3),     \
              "=&v"(_b3)                                                                              \
            : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "i"(mask_lo)   \
            : "vcc", "scc");                                                                          \
        a0 = _a0; b0 = _b0; a1 = _a1; b1 = _b1; a2 = _a2; b2 = _b2; a3 = _a3; b3 = _b3;               \
    } while (0)

// A' = vcc(bit clear) ? A : dpp(B): the hi lanes read B from lane l - m  (CTRL_HI);
// B' = vcc(bit set)   ? B : dpp(A): the lo lanes read A from lane l + m  (CTRL_LO).
template <int B>
__device__ __forceinline__ void xpose_round(float (&r)[32])
{
    constexpr int m = 1 << B;
    // lanes with bit B clear, per 32 lanes
    constexpr unsigned mask_lo = B == 0 ? 0x55555555u : B == 1 ? 0x33333333u : B == 2 ? 0x0f0f0f0fu : 0x00ff00ffu;
    // the 16 pairs (i, i | m), four per block
    int idx[16];
    {
        int n = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i)
            if (!(i & m)) idx[n++] = i;
    }
#pragma unroll
    for (int q = 0; q < 16; q += 4) {
        const int i0 = idx[q], i1 = idx[q + 1], i2 = idx[q + 2], i3 = idx[q + 3];
        if constexpr (B == 0)
            AF_XCHG4("quad_perm:[1,0,3,2]", "quad_perm:[1,0,3,2]", r[i0], r[i0 | m], r[i1], r[i1 | m], r[i2],
                     r[i2 | m], r[i3], r[i3 | m], mask_lo);
        else if constexpr (B == 1)
            AF_XCHG4("quad_perm:[2,3,0,1]", "quad_perm:[2,3,0,1]", r[i0], r[i0 | m], r[i1], r[i1 | m], r[i2],
                     r[i2 | m], r[i3], r[i3 | m], mask_lo);
        else if constexpr (B == 2)
            AF_XCHG4("row_shr:4", "row_shl:4", r[i0], r[i0 | m], r[i1], r[i1 | m], r[i2], r[i2 | m], r[i3],
                     r[i3 | m], mask_lo);
        else
            AF_XCHG4("row_ror:8", "row_ror:8", r[i0], r[i0 | m], r[i1], r[i1 | m], r[i2], r[i2 | m], r[i3],
                     r[i3 | m], mask_lo);
    }
}

// (register i, lane l) <-> (register l, lane i) within each half-wave
__device__ __forceinline__ void xpose32(float (&r)[32], int lane)
{
    xpose_round<0>(r);
    xpose_round<1>(r);
    xpose_round<2>(r);
    xpose_round<3>(r);
    // bit 4: odd rows of A <-> even rows of B, four pairs per block (the s_nop covers the
    // VALU-write -> permlane-read wait states that hipcc cannot see inside the asm)
#pragma unroll
    for (int i = 0; i < 16; i += 4)
        asm volatile(
            "s_nop 1\n\t"
            "v_permlane16_swap_b32 %0, %1\n\t"
            "v_permlane16_swap_b32 %2, %3\n\t"
            "v_permlane16_swap_b32 %4, %5\n\t"
            "v_permlane16_swap_b32 %6, %7"
            : "+v"(r[i]), "+v"(r[i | 16]), "+v"(r[i + 1]), "+v"(r[(i + 1) | 16]), "+v"(r[i + 2]),
              "+v"(r[(i + 2) | 16]), "+v"(r[i + 3]), "+v"(r[(i + 3) | 16]));
    (void)lane;
}

}  // namespace feat
}  // namespace af
