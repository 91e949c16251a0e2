/*
 * unet_tiff.h -- C ABI of libunet_tiff.so: the HOST-ONLY strip / tile decoders of the GeoTIFF reader (unet_amd/tiffio.py).
 * No device code, no HIP runtime: g++ links it (unet_amd/build.py: build_host_codecs), so a tile-preparation box without a GPU stack
 * reads the rasters GDAL writes.  The reference reads every raster through GDAL / rasterio (create_tiles_unet.py:252-434, data.py:18-28,
 * predict.py:206-222), which decode these compressions transparently.
 *
 * unet_tiff_lzw_decode / unet_tiff_packbits_decode are declared in unet_hip.h (libunet_hip.so exports them too); this header adds the
 * entry point that only the host library carries.
 */
#ifndef UNET_TIFF_H
#define UNET_TIFF_H

#ifdef __cplusplus
extern "C" {
#endif

/* JPEG-in-TIFF (Compression 7, TIFF Technical Note 2): one strip / tile is an abbreviated JPEG stream; `tables` (may be NULL) is the
 * JPEGTables tag 347 (SOI, DQT / DHT segments, EOI) read before it.  Baseline / extended sequential Huffman, 8-bit samples, 1-4 components,
 * restart intervals, one interleaved scan or one scan per component; chroma at full resolution or subsampled 2:1 horizontally (h2v1) or in
 * both directions (h2v2).  Arithmetic restated from the Independent JPEG Group's library as libjpeg-turbo ships it (jidctint.c "islow" IDCT,
 * jdsample.c triangle-filter "fancy" upsampling, jdcolor.c YCbCr -> RGB tables), because that is what libtiff -- and so GDAL -- calls: the
 * bytes equal theirs (tests/test_tiff_jpeg_cpu.py compares with libtiff's own decode through Pillow).
 *   ycbcr_to_rgb != 0: PhotometricInterpretation 6 -- three components are converted to RGB (what GDAL / Pillow ask libtiff for:
 *                      JPEGCOLORMODE_RGB); 0: components are stored as decoded (RGB, grey, 4-band imagery).
 *   dst[cap]: pixel-interleaved rows [rows][cols][comps]; dims[3] receives rows, cols, comps of the stream's own frame header.
 * Returns the number of bytes written (rows * cols * comps), -1 for a corrupt / truncated stream or a frame larger than cap,
 * -2 for a JPEG process this decoder does not implement (progressive, lossless, arithmetic coding, 12-bit, other sampling ratios). */
long long unet_tiff_jpeg_decode(const unsigned char* tables, long long ntables, const unsigned char* src, long long n, int ycbcr_to_rgb,
                                unsigned char* dst, long long cap, int* dims);

#ifdef __cplusplus
}
#endif
#endif
