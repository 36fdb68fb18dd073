// host_png.hip -- HOST code: PNG scanline reconstruction (PNG specification, section 9 "Filtering").
//
// The reference reads KITTI's 16-bit flow maps with PyPNG (flowutils/flow_io.py:104-127, `png.Reader`);
// Pillow cannot return 16-bit RGB samples.  kitti_io.py parses the chunks and inflates IDAT with zlib; the
// per-byte Sub / Up / Average / Paeth recurrences are sequential along and across scanlines, so they run
// here, on one host thread, in place.  Nothing in this file touches the GPU.
#include <cstdint>
#include <cstdlib>

#include "ufr_common.h"

// data: `rows` scanlines of (1 filter byte + stride payload bytes).  out: rows * stride bytes.
extern "C" int ufr_host_png_unfilter(const unsigned char* data, unsigned char* out, int rows, int stride, int bpp) {
  UFR_REQUIRE(data && out, "png unfilter: null pointer");
  UFR_REQUIRE(rows > 0 && stride > 0 && bpp > 0 && bpp <= 8 && stride % bpp == 0, "png unfilter: bad geometry");
  for (int r = 0; r < rows; ++r) {
    const uint8_t* in = data + (size_t)r * (stride + 1);
    const int ft = in[0];
    const uint8_t* x = in + 1;
    uint8_t* cur = out + (size_t)r * stride;
    const uint8_t* up = r > 0 ? cur - stride : nullptr;
    switch (ft) {
      case 0:
        for (int i = 0; i < stride; ++i) cur[i] = x[i];
        break;
      case 1:
        for (int i = 0; i < stride; ++i) cur[i] = (uint8_t)(x[i] + (i >= bpp ? cur[i - bpp] : 0));
        break;
      case 2:
        for (int i = 0; i < stride; ++i) cur[i] = (uint8_t)(x[i] + (up ? up[i] : 0));
        break;
      case 3:
        for (int i = 0; i < stride; ++i) {
          const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0;
          cur[i] = (uint8_t)(x[i] + ((a + b) >> 1));
        }
        break;
      case 4:
        for (int i = 0; i < stride; ++i) {
          const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
          const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
          const int pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          cur[i] = (uint8_t)(x[i] + pred);
        }
        break;
      default:
        return ufr::fail(UFR_EINVAL, "png unfilter: scanline %d has filter type %d", r, ft);
    }
  }
  return UFR_OK;
}
