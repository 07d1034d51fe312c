// MFMA operand fetch from a PIXEL-MAJOR LDS tile through the gfx950 transposing read.
#pragma once
#include "common.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// tile: bf16 [row][LD] with the MFMA reduction index along ROWS.  Returns the 32(col) x 16(row) operand of
// v_mfma_f32_32x32x16_bf16: lane L -> column col0 + (L&31), rows row0 + 8*(L>>5) .. +7.
// ds_read_b64_tr_b16 (probed on the box, tools/probes/tr_probe.hip): inside each 16-lane group, lane s fetches the 8
// bytes it addresses and lane i receives element (i%4) of source lanes 4j + i/4 (j = 0..3); addressing lane s at
// row s/4, columns 4*(s%4).. gives lane i the four rows of column i.
__device__ __forceinline__ bf16x8 tr_frag_bf16(const bf16_t* tile, int LD, int row0, int col0, int lane) {
    const int g = lane >> 4, s = lane & 15;
    const bf16_t* p = tile + (row0 + 8 * (g >> 1) + (s >> 2)) * LD + col0 + 16 * (g & 1) + 4 * (s & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 4 * LD));
    union { struct { s16x4 a, b; } h; bf16x8 f; } u;
    u.h.a = lo; u.h.b = hi;
    return u.f;
}
