// 8-point DFT on complex doubles in registers (decimation in frequency, three radix-2 stages with the W8 constants folded
// in).  Outputs land in bit-reversed slots: slot s holds y[kFftR8Slot[s]].  Shared by the ultrasonic power kernel
// (kernels.hip) and the block-cooperative STFT (stft.hip).
#pragma once

namespace bnhip {

__device__ __forceinline__ void fft_r8_dft8(double (&re)[8], double (&im)[8]) {
    constexpr double R = 0.70710678118654752440;
#pragma unroll
    for (int m = 0; m < 4; m++) {                    // stage 1: (m, m + 4), lower half times W8^m
        const double ur = re[m] + re[m + 4], ui = im[m] + im[m + 4], vr = re[m] - re[m + 4], vi = im[m] - im[m + 4];
        re[m] = ur; im[m] = ui;
        if (m == 0) { re[4] = vr; im[4] = vi; }
        else if (m == 1) { re[5] = (vr + vi) * R; im[5] = (vi - vr) * R; }       // (1 - i) / sqrt 2
        else if (m == 2) { re[6] = vi; im[6] = -vr; }                             // -i
        else { re[7] = (vi - vr) * R; im[7] = -(vr + vi) * R; }                   // (-1 - i) / sqrt 2
    }
#pragma unroll
    for (int h = 0; h < 8; h += 4)                   // stage 2: (m, m + 2) inside each half, lower element times W4^m
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const int a = h + m, b = h + m + 2;
            const double ur = re[a] + re[b], ui = im[a] + im[b], vr = re[a] - re[b], vi = im[a] - im[b];
            re[a] = ur; im[a] = ui;
            if (m == 0) { re[b] = vr; im[b] = vi; } else { re[b] = vi; im[b] = -vr; }
        }
#pragma unroll
    for (int a = 0; a < 8; a += 2) {                 // stage 3: (m, m + 1)
        const double ur = re[a] + re[a + 1], ui = im[a] + im[a + 1], vr = re[a] - re[a + 1], vi = im[a] - im[a + 1];
        re[a] = ur; im[a] = ui; re[a + 1] = vr; im[a + 1] = vi;
    }
}
__device__ constexpr int kFftR8Slot[8] = {0, 4, 2, 6, 1, 5, 3, 7};

}  // namespace bnhip
