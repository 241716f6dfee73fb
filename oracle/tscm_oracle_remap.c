/*
 * tscm_oracle_remap.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE), application of remap tables.
 *
 * cv::remap(src, dst, mapx, mapy, cv::INTER_LINEAR) as the reference calls it (TS.cpp:304, :329; border mode and
 * value defaulted: BORDER_CONSTANT, 0) for 8-bit images of 1 or 3 channels, and cv::cvtColor(BGR2GRAY) on the result
 * (findCorner.cpp:9-10).  OpenCV is an external dependency of the reference (CMakeLists.txt:6); the algorithm is
 * restated from its published implementation (modules/imgproc/src/imgwarp.cpp, remapBilinear / initInterTab2D;
 * color_yuv: BGR2GRAY for 8U):
 *   - the float map coordinates are converted to fixed point with 5 fractional bits: s = cvRound(m * 32) (float
 *     product, round half to even), integer part s >> 5, fraction s & 31;
 *   - the four weights are shorts with scale 2^15: saturate_cast<short>((1 - fy)(1 - fx) * 32768), ... -- the products
 *     are exact multiples of 32, so the weights sum to 32768 except at fx = fy = 0, where 32768 saturates to 32767
 *     and the table's sum correction gives the missing unit to another tap (no effect after the final rounding);
 *   - pixel = (sum w_i p_i + 2^14) >> 15; neighbours outside the image contribute the border value 0;
 *   - grey = (B * 1868 + G * 9617 + R * 4899 + 2^13) >> 14.
 * PARITY UNPINNED (no OpenCV here; see tscm_oracle.h).
 */
#include <math.h>
#include <stddef.h>

#include "tscm_oracle.h"

static int cv_round_f32(float v) { return (int)lrintf(v); }          /* cvRound: nearest, ties to even */

/* dst: map_h x map_w x (to_gray ? 1 : channels) */
void orc_remap_bilinear(const unsigned char *src, int w, int h, int stride, int channels, const float *mapx, const float *mapy, int map_w, int map_h,
                        int map_stride, int to_gray, unsigned char *dst, int dst_stride)
{
    for (int i = 0; i < map_h; ++i)
        for (int j = 0; j < map_w; ++j) {
            const int sx = cv_round_f32(mapx[(size_t)i * map_stride + j] * 32.0f), sy = cv_round_f32(mapy[(size_t)i * map_stride + j] * 32.0f);
            int ix = sx >> 5, iy = sy >> 5;
            ix = ix < -32768 ? -32768 : (ix > 32767 ? 32767 : ix);          /* saturate_cast<short> */
            iy = iy < -32768 ? -32768 : (iy > 32767 ? 32767 : iy);
            const int fx = sx & 31, fy = sy & 31;
            int wgt[4] = { 32 * (32 - fx) * (32 - fy), 32 * fx * (32 - fy), 32 * (32 - fx) * fy, 32 * fx * fy };
            if (wgt[0] == 32768) { wgt[0] = 32767; wgt[3] = 1; }
            int px[3] = { 0, 0, 0 };
            for (int c = 0; c < channels; ++c) {
                int acc = 0;
                for (int k = 0; k < 4; ++k) {
                    const int x = ix + (k & 1), y = iy + (k >> 1);
                    const int p = (x >= 0 && x < w && y >= 0 && y < h) ? src[(size_t)y * stride + (size_t)x * channels + c] : 0;
                    acc += wgt[k] * p;
                }
                int v = (acc + (1 << 14)) >> 15;
                v = v < 0 ? 0 : (v > 255 ? 255 : v);
                px[c] = v;
            }
            if (to_gray && channels == 3) dst[(size_t)i * dst_stride + j] = (unsigned char)((px[0] * 1868 + px[1] * 9617 + px[2] * 4899 + (1 << 13)) >> 14);
            else for (int c = 0; c < channels; ++c) dst[(size_t)i * dst_stride + (size_t)j * channels + c] = (unsigned char)px[c];
        }
}
