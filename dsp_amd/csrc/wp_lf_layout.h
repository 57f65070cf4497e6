// wp_lf_layout.h — where the points of a long-frame convolution's frame lie in the workspace A.
// Plain constexpr C++ (no HIP types): shared by the kernels and the host-side layout check
// (tests/host_emu/fft_emu.cpp).
#pragma once

namespace wp {

constexpr int LF_C = 2048;      // columns of a frame: Nc = R * C complex points, C = LF_C ...
constexpr int LF_C_BIG = 4096;  // ... or LF_C_BIG (frames of 2^21 samples: 256 rows of 4096 points)

// A frame's R x C points are stored in blocks of LF_RB = 32 rows x chunks of LF_CC = 64 columns,
// [row block][chunk][row][64 columns].  A row (what the row pass streams) is then 32 pieces of 512 bytes 16 KB
// apart, which spreads one workgroup's accesses over the memory channels, and the 256-byte runs of the column
// passes' tiles (32 columns of every row) sit 512 bytes apart instead of 16 KB.
#ifndef WP_LF_RB
#define WP_LF_RB 32  // rows per block (a multiple of 16)
#endif
#ifndef WP_LF_CC
#define WP_LF_CC 64  // columns per chunk (divides 128)
#endif
constexpr int LF_RB = WP_LF_RB, LF_CC = WP_LF_CC;
static_assert(LF_RB % 16 == 0 && 128 % LF_CC == 0 && LF_CC >= 32, "frame layout");
// point (k1, c) of a frame of C columns: lf_row_off(k1, C) + lf_col_off(c)
constexpr int lf_row_off(int k1, int C) { return (k1 / LF_RB) * (LF_RB * C) + (k1 % LF_RB) * LF_CC; }
constexpr int lf_col_off(int c) { return (c / LF_CC) * (LF_RB * LF_CC) + (c % LF_CC); }

}  // namespace wp
