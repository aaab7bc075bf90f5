/* TEST INFRASTRUCTURE: an INDEPENDENT TIFF codec (libtiff 4.2 from /opt/conda) on the other side of taudem_amd's own GeoTIFF
 * reader / writer (taudem_amd/csrc/geotiff.cpp), which otherwise would only ever be tested against itself (the oracle's GDAL
 * shim reads through the same code).
 *   tiffx write <file> <kind> <nx> <ny>   kind: f32_tiled_deflate_pred3 | i16_strip_lzw_pred2 | i32_tiled_none | f32_strip_packbits |
 *                                         f32_big_tiled_deflate (BigTIFF) | f32_strip_lzw
 *                                         pixels: value(x, y) below, so the test can regenerate them
 *   tiffx dump  <file> <out.raw>           decodes with libtiff (scanline / tile API) and writes the raw native pixels
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <tiffio.h>

static double value(long x, long y) { return (double)((x * 31 + y * 17) % 1000) * 0.25 - 50.0 + (double)((x ^ y) & 7); }

static int do_write(const char* path, const char* kind, long nx, long ny) {
    const int big = strstr(kind, "big") != NULL;
    TIFF* t = TIFFOpen(path, big ? "w8" : "w");
    if (!t) return 2;
    int fmt = SAMPLEFORMAT_IEEEFP, bits = 32;
    if (!strncmp(kind, "i16", 3)) { fmt = SAMPLEFORMAT_INT; bits = 16; }
    if (!strncmp(kind, "i32", 3)) { fmt = SAMPLEFORMAT_INT; bits = 32; }
    int comp = COMPRESSION_NONE, pred = 1;
    if (strstr(kind, "deflate")) comp = COMPRESSION_ADOBE_DEFLATE;
    if (strstr(kind, "lzw")) comp = COMPRESSION_LZW;
    if (strstr(kind, "packbits")) comp = COMPRESSION_PACKBITS;
    if (strstr(kind, "pred3")) pred = 3;
    if (strstr(kind, "pred2")) pred = 2;
    const int tiled = strstr(kind, "tiled") != NULL;
    TIFFSetField(t, TIFFTAG_IMAGEWIDTH, (uint32_t)nx);
    TIFFSetField(t, TIFFTAG_IMAGELENGTH, (uint32_t)ny);
    TIFFSetField(t, TIFFTAG_SAMPLESPERPIXEL, 1);
    TIFFSetField(t, TIFFTAG_BITSPERSAMPLE, bits);
    TIFFSetField(t, TIFFTAG_SAMPLEFORMAT, fmt);
    TIFFSetField(t, TIFFTAG_PLANARCONFIG, PLANARCONFIG_CONTIG);
    TIFFSetField(t, TIFFTAG_PHOTOMETRIC, PHOTOMETRIC_MINISBLACK);
    TIFFSetField(t, TIFFTAG_COMPRESSION, comp);
    if (pred != 1) TIFFSetField(t, TIFFTAG_PREDICTOR, pred);
    const size_t bps = (size_t)bits / 8;
    if (tiled) {
        const uint32_t tw = 64, th = 48;
        TIFFSetField(t, TIFFTAG_TILEWIDTH, tw);
        TIFFSetField(t, TIFFTAG_TILELENGTH, th);
        unsigned char* buf = (unsigned char*)malloc((size_t)tw * th * bps);
        for (long y0 = 0; y0 < ny; y0 += th)
            for (long x0 = 0; x0 < nx; x0 += tw) {
                for (uint32_t j = 0; j < th; j++)
                    for (uint32_t i = 0; i < tw; i++) {
                        const double v = value(x0 + i, y0 + j);   /* padding beyond the image is arbitrary */
                        unsigned char* q = buf + ((size_t)j * tw + i) * bps;
                        if (fmt == SAMPLEFORMAT_IEEEFP) { float f = (float)v; memcpy(q, &f, 4); }
                        else if (bits == 16) { int16_t s = (int16_t)v; memcpy(q, &s, 2); }
                        else { int32_t s = (int32_t)v; memcpy(q, &s, 4); }
                    }
                if (TIFFWriteTile(t, buf, (uint32_t)x0, (uint32_t)y0, 0, 0) < 0) return 3;
            }
        free(buf);
    } else {
        TIFFSetField(t, TIFFTAG_ROWSPERSTRIP, 7);
        unsigned char* row = (unsigned char*)malloc((size_t)nx * bps);
        for (long y = 0; y < ny; y++) {
            for (long x = 0; x < nx; x++) {
                const double v = value(x, y);
                unsigned char* q = row + (size_t)x * bps;
                if (fmt == SAMPLEFORMAT_IEEEFP) { float f = (float)v; memcpy(q, &f, 4); }
                else if (bits == 16) { int16_t s = (int16_t)v; memcpy(q, &s, 2); }
                else { int32_t s = (int32_t)v; memcpy(q, &s, 4); }
            }
            if (TIFFWriteScanline(t, row, (uint32_t)y, 0) < 0) return 3;
        }
        free(row);
    }
    TIFFClose(t);
    return 0;
}

static int do_dump(const char* path, const char* out) {
    TIFF* t = TIFFOpen(path, "r");
    if (!t) return 2;
    uint32_t nx = 0, ny = 0; uint16_t bits = 0;
    TIFFGetField(t, TIFFTAG_IMAGEWIDTH, &nx);
    TIFFGetField(t, TIFFTAG_IMAGELENGTH, &ny);
    TIFFGetField(t, TIFFTAG_BITSPERSAMPLE, &bits);
    const size_t bps = bits / 8;
    FILE* f = fopen(out, "wb");
    if (!f) return 4;
    if (TIFFIsTiled(t)) {
        uint32_t tw = 0, th = 0;
        TIFFGetField(t, TIFFTAG_TILEWIDTH, &tw);
        TIFFGetField(t, TIFFTAG_TILELENGTH, &th);
        unsigned char* img = (unsigned char*)malloc((size_t)nx * ny * bps);
        unsigned char* buf = (unsigned char*)malloc((size_t)TIFFTileSize(t));
        for (uint32_t y0 = 0; y0 < ny; y0 += th)
            for (uint32_t x0 = 0; x0 < nx; x0 += tw) {
                if (TIFFReadTile(t, buf, x0, y0, 0, 0) < 0) return 3;
                for (uint32_t j = 0; j < th && y0 + j < ny; j++) {
                    const uint32_t w = x0 + tw <= nx ? tw : nx - x0;
                    memcpy(img + ((size_t)(y0 + j) * nx + x0) * bps, buf + (size_t)j * tw * bps, (size_t)w * bps);
                }
            }
        fwrite(img, bps, (size_t)nx * ny, f);
        free(img); free(buf);
    } else {
        unsigned char* row = (unsigned char*)malloc((size_t)TIFFScanlineSize(t));
        for (uint32_t y = 0; y < ny; y++) {
            if (TIFFReadScanline(t, row, y, 0) < 0) return 3;
            fwrite(row, bps, nx, f);
        }
        free(row);
    }
    fclose(f);
    TIFFClose(t);
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 6 && !strcmp(argv[1], "write")) return do_write(argv[2], argv[3], atol(argv[4]), atol(argv[5]));
    if (argc >= 4 && !strcmp(argv[1], "dump")) return do_dump(argv[2], argv[3]);
    fprintf(stderr, "usage: tiffx write <file> <kind> <nx> <ny> | tiffx dump <file> <out.raw>\n");
    return 1;
}
