// npyio.h -- minimal NumPy .npy reader/writer for the `mgm` host program.
// The reference reads and writes images through the vendored iio library (img_tools.h:18-34),
// which understands NPY v1.0 natively (iio.c:3178-3258, 4269); this is the only container both
// programs share without libpng/libtiff.  Images are returned in the reference's planar `Img`
// layout data[x + y*nx + c*nx*ny] (img.h:35-51); files hold (h,w) or (h,w,c) arrays.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

struct HostImg {
    std::vector<float> data;
    int nx = 0, ny = 0, nch = 0;
    int npix() const { return nx * ny; }
};

namespace npy {

inline HostImg read(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    unsigned char magic[10];
    if (fread(magic, 1, 10, f) != 10 || memcmp(magic, "\x93NUMPY", 6)) {
        fclose(f);
        throw std::runtime_error(path + ": not an .npy file (only .npy images are supported by this build)");
    }
    size_t hlen = magic[8] | (magic[9] << 8);
    if (magic[6] >= 2) {  // v2/v3: 4-byte header length
        unsigned char more[2];
        if (fread(more, 1, 2, f) != 2) { fclose(f); throw std::runtime_error(path + ": truncated header"); }
        hlen |= (size_t)more[0] << 16 | (size_t)more[1] << 24;
    }
    std::string h(hlen, ' ');
    if (fread(&h[0], 1, hlen, f) != hlen) { fclose(f); throw std::runtime_error(path + ": truncated header"); }
    auto field = [&](const char *key) {
        size_t p = h.find(key);
        if (p == std::string::npos) throw std::runtime_error(path + ": missing " + key);
        return h.substr(h.find(':', p) + 1);
    };
    std::string descr = field("'descr'");
    descr = descr.substr(descr.find('\'') + 1);
    descr = descr.substr(0, descr.find('\''));
    // fortran_order: True -- iio swaps the two image sides, reads, and transposes (iio.c:3209-3252): for a 2-D array exactly numpy's
    // column-major layout (np.save of an F-contiguous array, e.g. a transposed view); with a third axis iio still treats it as
    // the interleaved pixel dimension, and so does this reader (same bytes, same image)
    const bool fortran = field("'fortran_order'").find("True") < 8;
    std::string sh = field("'shape'");
    sh = sh.substr(sh.find('(') + 1);
    sh = sh.substr(0, sh.find(')'));
    std::vector<long> dims;
    for (size_t p = 0; p < sh.size();) {
        while (p < sh.size() && (sh[p] == ' ' || sh[p] == ',')) p++;
        if (p >= sh.size()) break;
        dims.push_back(strtol(sh.c_str() + p, nullptr, 10));
        while (p < sh.size() && sh[p] != ',') p++;
    }
    if (dims.size() < 2 || dims.size() > 3) { fclose(f); throw std::runtime_error(path + ": expected (h,w) or (h,w,c)"); }
    if (dims.size() == 3 && dims[0] == 1 && dims[1] > 1 && dims[2] > 1) dims = {dims[1], dims[2]};  // iio's "squeeze" (iio.c:3202-3207)
    HostImg im;
    im.ny = (int)dims[0];
    im.nx = (int)dims[1];
    im.nch = dims.size() == 3 ? (int)dims[2] : 1;
    const size_t n = (size_t)im.nx * im.ny * im.nch;
    std::vector<float> inter(n);
    auto rd = [&](size_t esz) {
        std::vector<unsigned char> raw(n * esz);
        if (fread(raw.data(), esz, n, f) != n) { fclose(f); throw std::runtime_error(path + ": truncated data"); }
        return raw;
    };
    if (descr == "<f4" || descr == "=f4" || descr == "|f4") {
        if (fread(inter.data(), 4, n, f) != n) { fclose(f); throw std::runtime_error(path + ": truncated data"); }
    } else if (descr == "<f8") {
        auto raw = rd(8);
        for (size_t i = 0; i < n; i++) { double d; memcpy(&d, &raw[i * 8], 8); inter[i] = (float)d; }
    } else if (descr == "|u1") {
        auto raw = rd(1);
        for (size_t i = 0; i < n; i++) inter[i] = raw[i];
    } else if (descr == "<u2") {
        auto raw = rd(2);
        for (size_t i = 0; i < n; i++) { uint16_t v; memcpy(&v, &raw[i * 2], 2); inter[i] = v; }
    } else if (descr == "<i4") {
        auto raw = rd(4);
        for (size_t i = 0; i < n; i++) { int32_t v; memcpy(&v, &raw[i * 4], 4); inter[i] = (float)v; }
    } else {
        fclose(f);
        throw std::runtime_error(path + ": unsupported dtype " + descr);
    }
    fclose(f);
    // interleaved (h,w,c) -> planar, as iio_read_image_float_split does
    im.data.resize(n);
    const size_t np = (size_t)im.nx * im.ny;
    if (fortran) {  // the file holds the transposed image: pixel (x, y) sits at (x * ny + y)
        for (int y = 0; y < im.ny; y++)
            for (int x = 0; x < im.nx; x++)
                for (int c = 0; c < im.nch; c++) im.data[(size_t)y * im.nx + x + c * np] = inter[((size_t)x * im.ny + y) * im.nch + c];
        return im;
    }
    for (size_t p = 0; p < np; p++)
        for (int c = 0; c < im.nch; c++) im.data[p + c * np] = inter[p * im.nch + c];
    return im;
}

inline void write(const std::string &path, const HostImg &im)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    char dict[128];
    snprintf(dict, sizeof dict, "{'descr': '<f4', 'fortran_order': False, 'shape': (%d, %d, %d), }", im.ny, im.nx, im.nch);
    std::string h(dict);
    while ((10 + h.size() + 1) % 64) h += ' ';
    h += '\n';
    const unsigned char magic[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char)(h.size() & 255),
                                     (unsigned char)(h.size() >> 8)};
    fwrite(magic, 1, 10, f);
    fwrite(h.data(), 1, h.size(), f);
    const size_t np = (size_t)im.nx * im.ny;
    std::vector<float> inter(np * im.nch);
    for (size_t p = 0; p < np; p++)
        for (int c = 0; c < im.nch; c++) inter[p * im.nch + c] = im.data[p + c * np];
    fwrite(inter.data(), 4, inter.size(), f);
    fclose(f);
}

}  // namespace npy
