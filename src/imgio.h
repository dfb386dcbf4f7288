// imgio.h -- image files for the `mgm` host program: PNG, TIFF, PGM/PPM, PFM and .npy in; float TIFF, PFM and
// .npy out.  The reference goes through its vendored iio library on top of libpng/libtiff (img_tools.h:18-34,
// iio/iio.c); this build has neither library's headers, so the decoders are written here against the format
// specifications (PNG 1.2 incl. Adam7, TIFF 6.0 + BigTIFF, Netpbm) with zlib as the only dependency.  What a file decodes TO
// follows iio: samples become floats unchanged (8/16-bit unsigned, float), PNG sub-byte grey is scaled to 0..255,
// palettes become RGB, tRNS becomes an alpha channel (iio.c:1501-1504: PACKING | EXPAND), TIFF photometric
// interpretation is ignored (iio reads raw scanlines, iio.c:1657-1880), and images come back in the planar `Img`
// layout data[x + y*nx + c*nx*ny] (img.h:35-51).  Outputs: the reference writes float TIFFs (LZW below 2000x2000,
// iio.c:4008-4011); this writer emits the same samples as an uncompressed float32 TIFF every TIFF reader accepts.
#pragma once
#include <zlib.h>

#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "npyio.h"

namespace imgio {

using bytes = std::vector<uint8_t>;

inline bytes slurp(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    bytes b;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
    fclose(f);
    return b;
}

inline void inflate_into(const uint8_t *src, size_t n, uint8_t *dst, size_t expect, const std::string &what)
{
    z_stream z{};
    if (inflateInit(&z) != Z_OK) throw std::runtime_error(what + ": zlib init failed");
    z.next_in = const_cast<Bytef *>(src);
    z.next_out = dst;
    size_t in_left = n, out_left = expect;
    int r = Z_OK;
    while (r == Z_OK && out_left) {  // feed in < 4 GiB pieces (uInt counters)
        const uInt ci = (uInt)std::min<size_t>(in_left, 1u << 30), co = (uInt)std::min<size_t>(out_left, 1u << 30);
        z.avail_in = ci;
        z.avail_out = co;
        r = inflate(&z, Z_NO_FLUSH);
        in_left -= ci - z.avail_in;
        out_left -= co - z.avail_out;
        if (r == Z_BUF_ERROR && in_left && out_left) r = Z_OK;
    }
    inflateEnd(&z);
    if (out_left) throw std::runtime_error(what + ": compressed data ends early or is corrupt");
}

// A corrupt header must not turn into a huge allocation: no supported coding expands by more than ~1:4096.
inline bool plausible(double decoded_bytes, size_t coded_bytes) { return decoded_bytes <= (double)coded_bytes * 4096.0 + 65536.0; }

// planar float image from interleaved samples
template <class GET>
inline HostImg planar(int nx, int ny, int nch, GET get)
{
    HostImg im;
    im.nx = nx;
    im.ny = ny;
    im.nch = nch;
    const size_t np = (size_t)nx * ny;
    im.data.resize(np * nch);
    for (size_t p = 0; p < np; p++)
        for (int c = 0; c < nch; c++) im.data[p + c * np] = get(p * nch + c);
    return im;
}

// ------------------------------------------------------------------------------------------------ PNG
namespace png {

inline uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

inline HostImg decode(const bytes &f, const std::string &path)
{
    auto bad = [&](const char *m) { return std::runtime_error(path + ": " + m); };
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    bytes idat, plte, trns;
    bool end = false;
    while (!end && pos + 12 <= f.size()) {
        const uint32_t len = be32(&f[pos]);
        const char *type = (const char *)&f[pos + 4];
        if (pos + 12 + (size_t)len > f.size()) throw bad("truncated chunk");
        const uint8_t *d = &f[pos + 8];
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) throw bad("bad IHDR");
            w = be32(d);
            h = be32(d + 4);
            depth = d[8];
            ctype = d[9];
            interlace = d[12];
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(d, d + len);
        } else if (!memcmp(type, "tRNS", 4)) {
            trns.assign(d, d + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), d, d + len);
        } else if (!memcmp(type, "IEND", 4)) {
            end = true;
        }
        pos += 12 + (size_t)len;
    }
    int ch;
    switch (ctype) {
    case 0: ch = 1; break;
    case 2: ch = 3; break;
    case 3: ch = 1; break;
    case 4: ch = 2; break;
    case 6: ch = 4; break;
    default: throw bad("bad colour type");
    }
    if (!w || !h || idat.empty()) throw bad("no image data");
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4))) ||
        (ctype == 3 && depth == 16))
        throw bad("bad bit depth");
    if (interlace > 1) throw bad("unknown interlace method");
    // One pass for a plain file, seven reduced images for Adam7 (PNG 1.2 section 8.2): each pass is a complete
    // filtered image of the pixels x0 + k dx, y0 + l dy.
    struct Pass { uint32_t x0, y0, dx, dy; };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass whole[1] = {{0, 0, 1, 1}};
    const Pass *passes = interlace ? adam7 : whole;
    const int npass = interlace ? 7 : 1;
    auto pass_w = [&](const Pass &q) { return q.x0 < w ? (w - q.x0 + q.dx - 1) / q.dx : 0u; };
    auto pass_h = [&](const Pass &q) { return q.y0 < h ? (h - q.y0 + q.dy - 1) / q.dy : 0u; };
    auto row_bytes = [&](uint32_t pw) { return ((size_t)pw * ch * depth + 7) / 8; };
    double total = 0;
    for (int k = 0; k < npass; k++)
        if (pass_w(passes[k]) && pass_h(passes[k])) total += (double)(row_bytes(pass_w(passes[k])) + 1) * pass_h(passes[k]);
    if (!plausible(total, idat.size()) || !plausible((double)w * h * ch * 4, idat.size())) throw bad("image size does not fit the file");
    bytes raw((size_t)total);
    inflate_into(idat.data(), idat.size(), raw.data(), raw.size(), path);

    const size_t bpp = std::max<size_t>(1, (size_t)ch * depth / 8);
    std::vector<unsigned> smp((size_t)w * h * ch);  // samples as written (before any expansion)
    size_t off = 0;
    for (int k = 0; k < npass; k++) {
        const Pass &q = passes[k];
        const uint32_t pw = pass_w(q), ph = pass_h(q);
        if (!pw || !ph) continue;
        const size_t rowbytes = row_bytes(pw);
        bytes zero(rowbytes, 0);
        for (uint32_t y = 0; y < ph; y++) {
            // undo the scanline filter in place (PNG 1.2 section 6)
            uint8_t *cur = &raw[off + (rowbytes + 1) * y + 1];
            const uint8_t *up = y ? cur - (rowbytes + 1) : zero.data();
            switch (cur[-1]) {
            case 0: break;
            case 1:
                for (size_t i = bpp; i < rowbytes; i++) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
                break;
            case 2:
                for (size_t i = 0; i < rowbytes; i++) cur[i] = (uint8_t)(cur[i] + up[i]);
                break;
            case 3:
                for (size_t i = 0; i < rowbytes; i++) cur[i] = (uint8_t)(cur[i] + (((i >= bpp ? cur[i - bpp] : 0) + up[i]) >> 1));
                break;
            case 4:
                for (size_t i = 0; i < rowbytes; i++) {
                    const int a = i >= bpp ? cur[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
                    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                    cur[i] = (uint8_t)(cur[i] + (pa <= pb && pa <= pc ? a : pb <= pc ? b : c));
                }
                break;
            default: throw bad("bad filter type");
            }
            for (uint32_t x = 0; x < pw; x++)
                for (int c = 0; c < ch; c++) {
                    const size_t i = (size_t)x * ch + c;
                    unsigned v;
                    if (depth == 8) v = cur[i];
                    else if (depth == 16) v = (unsigned)cur[2 * i] << 8 | cur[2 * i + 1];
                    else {
                        const size_t bit = i * depth;
                        v = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
                    }
                    smp[((size_t)(q.y0 + y * q.dy) * w + (q.x0 + x * q.dx)) * ch + c] = v;
                }
        }
        off += (rowbytes + 1) * ph;
    }
    auto sample = [&](uint32_t y, size_t i) -> unsigned { return smp[(size_t)y * w * ch + i]; };  // i-th sample of row y
    const unsigned maxv = depth == 16 ? 65535u : 255u;
    const bool alpha = !trns.empty() && (ctype == 0 || ctype == 2 || ctype == 3);
    const int och = (ctype == 3 ? 3 : ch) + (alpha ? 1 : 0);
    std::vector<float> inter((size_t)w * h * och);
    unsigned key[3] = {0, 0, 0};
    if (alpha && ctype != 3) {
        const int nk = ctype == 0 ? 1 : 3;
        if ((int)trns.size() < 2 * nk) throw bad("bad tRNS");
        for (int k = 0; k < nk; k++) key[k] = (unsigned)trns[2 * k] << 8 | trns[2 * k + 1];
    }
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            float *o = &inter[((size_t)y * w + x) * och];
            if (ctype == 3) {
                const unsigned idx = sample(y, x);
                if ((size_t)idx * 3 + 2 >= plte.size()) throw bad("palette index out of range");
                o[0] = plte[idx * 3];
                o[1] = plte[idx * 3 + 1];
                o[2] = plte[idx * 3 + 2];
                if (alpha) o[3] = idx < trns.size() ? trns[idx] : 255;
            } else {
                bool iskey = alpha;
                for (int c = 0; c < ch; c++) {
                    const unsigned s = sample(y, (size_t)x * ch + c);
                    if (alpha && s != key[c]) iskey = false;
                    // sub-byte grey is scaled to the 8-bit range (png_set_expand_gray_1_2_4_to_8)
                    o[c] = depth < 8 ? (float)(s * (255u / ((1u << depth) - 1))) : (float)s;
                }
                if (alpha) o[ch] = iskey ? 0.f : (float)maxv;
            }
        }
    return planar((int)w, (int)h, och, [&](size_t i) { return inter[i]; });
}

}  // namespace png

// ------------------------------------------------------------------------------------------------ TIFF
namespace tiff {

struct Reader {
    const bytes &f;
    const std::string &path;
    bool be = false, big = false;
    std::runtime_error bad(const std::string &m) const { return std::runtime_error(path + ": " + m); }
    uint64_t rd(size_t off, int n) const
    {
        if (off > f.size() || (size_t)n > f.size() - off) throw bad("truncated TIFF");
        uint64_t v = 0;
        for (int i = 0; i < n; i++) v |= (uint64_t)f[off + (be ? n - 1 - i : i)] << (8 * i);
        return v;
    }
};

// TIFF 6.0 section 13: MSB-first codes from 9 bits, ClearCode 256, EndOfInformation 257, width change one code early
inline void lzw(const uint8_t *src, size_t n, uint8_t *dst, size_t expect, const Reader &R)
{
    std::vector<int> prefix(4096, -1);
    std::vector<uint8_t> suffix(4096), first(4096);
    std::vector<uint32_t> length(4096, 1);
    for (int i = 0; i < 256; i++) suffix[i] = first[i] = (uint8_t)i;
    int bits = 9, next = 258, prev = -1;
    uint32_t acc = 0;
    int nacc = 0;
    size_t ip = 0, op = 0;
    auto emit = [&](int code) {  // write the string of `code` at op
        const uint32_t len = length[code];
        if (op + len > expect) {  // the last strip row may be cut short by the expected size
            std::vector<uint8_t> tmp(len);
            int c = code;
            for (uint32_t k = len; k-- > 0; c = prefix[c]) tmp[k] = suffix[c];
            memcpy(dst + op, tmp.data(), expect - op);
            op = expect;
            return;
        }
        int c = code;
        for (uint32_t k = len; k-- > 0; c = prefix[c]) dst[op + k] = suffix[c];
        op += len;
    };
    while (op < expect) {
        while (nacc < bits) {
            if (ip >= n) throw R.bad("LZW data ends early");
            acc = acc << 8 | src[ip++];
            nacc += 8;
        }
        const int code = (acc >> (nacc - bits)) & ((1 << bits) - 1);
        nacc -= bits;
        if (code == 257) break;
        if (code == 256) {
            bits = 9;
            next = 258;
            prev = -1;
            continue;
        }
        if (prev < 0) {
            if (code > 255) throw R.bad("corrupt LZW data");
            emit(code);
            prev = code;
            continue;
        }
        if (code > next || next >= 4096) throw R.bad("corrupt LZW data");
        prefix[next] = prev;
        first[next] = first[prev];
        length[next] = length[prev] + 1;
        suffix[next] = code < next ? first[code] : first[prev];
        next++;
        emit(code);
        if (next == (1 << bits) - 1 && bits < 12) bits++;
        prev = code;
    }
    if (op < expect) throw R.bad("LZW data ends early");
}

inline void packbits(const uint8_t *src, size_t n, uint8_t *dst, size_t expect, const Reader &R)
{
    size_t ip = 0, op = 0;
    while (op < expect && ip < n) {
        const int c = (int8_t)src[ip++];
        if (c >= 0) {
            const size_t k = std::min<size_t>(c + 1, expect - op);
            if (ip + k > n) throw R.bad("PackBits data ends early");
            memcpy(dst + op, src + ip, k);
            ip += c + 1;
            op += k;
        } else if (c != -128) {
            if (ip >= n) throw R.bad("PackBits data ends early");
            const size_t k = std::min<size_t>(1 - c, expect - op);
            memset(dst + op, src[ip++], k);
            op += k;
        }
    }
    if (op < expect) throw R.bad("PackBits data ends early");
}

inline HostImg decode(const bytes &f, const std::string &path)
{
    Reader R{f, path};
    if (f.size() < 8) throw R.bad("truncated TIFF");
    R.be = f[0] == 'M';
    const unsigned version = (unsigned)R.rd(2, 2);
    R.big = version == 43;
    if (version != 42 && version != 43) throw R.bad("not a TIFF file");
    size_t ifd = R.big ? (size_t)R.rd(8, 8) : (size_t)R.rd(4, 4);
    const size_t nent = R.big ? (size_t)R.rd(ifd, 8) : (size_t)R.rd(ifd, 2);
    const size_t ent0 = ifd + (R.big ? 8 : 2), entsz = R.big ? 20 : 12;
    static const int tsize[] = {0, 1, 1, 2, 4, 8, 1, 1, 2, 4, 8, 4, 8, 4, 0, 0, 8, 8, 8};
    auto values = [&](unsigned tag, std::vector<uint64_t> &out) {
        for (size_t e = 0; e < nent; e++) {
            const size_t p = ent0 + e * entsz;
            if ((unsigned)R.rd(p, 2) != tag) continue;
            const unsigned type = (unsigned)R.rd(p + 2, 2);
            const uint64_t cnt = R.big ? R.rd(p + 4, 8) : R.rd(p + 4, 4);
            if (type == 0 || type > 18 || !tsize[type]) throw R.bad("bad tag type");
            const int sz = tsize[type];
            if (cnt > f.size()) throw R.bad("bad tag count");
            const size_t inl = R.big ? 8 : 4, vp = p + (R.big ? 12 : 8);
            const size_t off = cnt * sz <= inl ? vp : (size_t)R.rd(vp, (int)inl);
            out.resize(cnt);
            for (uint64_t i = 0; i < cnt; i++) out[i] = R.rd(off + i * sz, sz);
            return true;
        }
        return false;
    };
    auto value = [&](unsigned tag, uint64_t dflt) {
        std::vector<uint64_t> v;
        return values(tag, v) && !v.empty() ? v[0] : dflt;
    };
    const uint64_t W = value(256, 0), H = value(257, 0);
    if (!W || !H) throw R.bad("TIFF without a size");
    const int spp = (int)value(277, 1), comp = (int)value(259, 1), planarcfg = (int)value(284, 1), pred = (int)value(317, 1);
    std::vector<uint64_t> v;
    int bps = 1;
    if (values(258, v)) {
        bps = (int)v[0];
        for (auto b : v)
            if ((int)b != bps) throw R.bad("mixed bits per sample");
    }
    int fmt = 1;
    if (values(339, v) && !v.empty()) fmt = (int)v[0];
    if (!(bps == 8 || bps == 16 || bps == 32 || bps == 64) || (fmt == 3 && bps < 32) || fmt < 1 || fmt > 3 ||
        (bps == 64 && fmt != 3))
        throw R.bad("unsupported sample type (" + std::to_string(bps) + " bits, format " + std::to_string(fmt) + ")");
    if (value(266, 1) != 1) throw R.bad("FillOrder 2 is not supported");
    const int bytesps = bps / 8;
    const bool tiled = value(322, 0) != 0;
    const uint64_t cw = tiled ? value(322, 0) : W;  // chunk = tile or strip
    uint64_t chh = tiled ? value(323, 0) : value(278, H);
    if (!tiled && chh > H) chh = H;
    if (!cw || !chh) throw R.bad("bad tile/strip size");
    std::vector<uint64_t> offs, cnts;
    if (!values(tiled ? 324 : 273, offs) || !values(tiled ? 325 : 279, cnts)) {
        if (!tiled && comp == 1 && offs.size() == 1) cnts.assign(1, f.size() - offs[0]);  // some writers omit the byte count
        else throw R.bad("missing strip/tile tables");
    }
    const uint64_t across = (W + cw - 1) / cw, down = (H + chh - 1) / chh;
    const int planes = planarcfg == 2 ? spp : 1, cs = planarcfg == 2 ? 1 : spp;  // samples per pixel within a chunk
    if (offs.size() < across * down * planes || cnts.size() < offs.size()) throw R.bad("short strip/tile tables");

    if (W > 0x7fffffff || H > 0x7fffffff || spp < 1 || spp > 1024 ||
        !plausible((double)W * (double)H * spp * bytesps, f.size()) || !plausible((double)cw * (double)chh * spp * bytesps, f.size()))
        throw R.bad("image size does not fit the file");
    const size_t np = (size_t)W * H;
    HostImg im;
    im.nx = (int)W;
    im.ny = (int)H;
    im.nch = spp;
    im.data.resize(np * spp);
    if (pred < 1 || pred > 3 || (pred == 2 && fmt == 3) || (pred == 3 && fmt != 3)) throw R.bad("unsupported TIFF predictor");
    auto raw = [&](const uint8_t *p) {  // one sample in file byte order
        uint64_t x = 0;
        for (int i = 0; i < bytesps; i++) x |= (uint64_t)p[R.be ? bytesps - 1 - i : i] << (8 * i);
        return x;
    };
    auto put = [&](uint8_t *p, uint64_t x) {
        for (int i = 0; i < bytesps; i++) p[R.be ? bytesps - 1 - i : i] = (uint8_t)(x >> (8 * i));
    };
    bytes buf, tmp;
    for (int pl = 0; pl < planes; pl++)
        for (uint64_t cy = 0; cy < down; cy++)
            for (uint64_t cx = 0; cx < across; cx++) {
                const size_t ci = (size_t)((pl * down + cy) * across + cx);
                const uint64_t rows = tiled ? chh : std::min<uint64_t>(chh, H - cy * chh);
                const size_t rowb = (size_t)cw * cs * bytesps, expect = rowb * rows;
                if (offs[ci] > f.size() || cnts[ci] > f.size() - offs[ci]) throw R.bad("strip/tile outside the file");
                const uint8_t *src = &f[offs[ci]];
                buf.resize(expect);
                switch (comp) {
                case 1:
                    if (cnts[ci] < expect) throw R.bad("short strip/tile");
                    memcpy(buf.data(), src, expect);
                    break;
                case 5: lzw(src, cnts[ci], buf.data(), expect, R); break;
                case 8:
                case 32946: inflate_into(src, cnts[ci], buf.data(), expect, path); break;
                case 32773: packbits(src, cnts[ci], buf.data(), expect, R); break;
                default: throw R.bad("unsupported TIFF compression " + std::to_string(comp));
                }
                const size_t ns = (size_t)cw * cs;  // samples per chunk row
                for (uint64_t r = 0; r < rows && pred != 1; r++) {
                    uint8_t *row = &buf[r * rowb];
                    if (pred == 3) {  // floating-point predictor (TIFF Technical Note 3): byte planes, MSB first
                        for (size_t i = cs; i < rowb; i++) row[i] = (uint8_t)(row[i] + row[i - cs]);
                        tmp.assign(row, row + rowb);
                        for (size_t i = 0; i < ns; i++)
                            for (int b = 0; b < bytesps; b++) row[i * bytesps + (R.be ? b : bytesps - 1 - b)] = tmp[b * ns + i];
                    } else {  // 2: horizontal differencing of the samples, modulo 2^bps
                        for (size_t i = cs; i < ns; i++) put(&row[i * bytesps], raw(&row[i * bytesps]) + raw(&row[(i - cs) * bytesps]));
                    }
                }
                for (uint64_t r = 0; r < rows; r++) {
                    const uint64_t y = cy * chh + r;
                    if (y >= H) break;
                    for (uint64_t xx = 0; xx < cw; xx++) {
                        const uint64_t x = cx * cw + xx;
                        if (x >= W) break;
                        for (int c = 0; c < cs; c++) {
                            const uint64_t b = raw(&buf[r * rowb + ((size_t)xx * cs + c) * bytesps]);
                            float val;
                            if (fmt == 3) {
                                if (bps == 32) {
                                    const uint32_t u = (uint32_t)b;
                                    memcpy(&val, &u, 4);
                                } else {
                                    double d;
                                    memcpy(&d, &b, 8);
                                    val = (float)d;
                                }
                            } else if (fmt == 2) {
                                val = bps == 8 ? (float)(int8_t)b : bps == 16 ? (float)(int16_t)b : (float)(int32_t)b;
                            } else {
                                val = bps == 32 ? (float)(uint32_t)b : (float)b;
                            }
                            im.data[(size_t)y * W + x + (size_t)(planarcfg == 2 ? pl : c) * np] = val;
                        }
                    }
                }
            }
    return im;
}

inline void write(const std::string &path, const HostImg &im)
{
    // little-endian classic TIFF, one strip, float32 samples, pixel-interleaved; BigTIFF above 4 GB
    const size_t np = (size_t)im.nx * im.ny, nbytes = np * im.nch * 4;
    const bool big = nbytes > 0xfff00000ull;
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    bytes h;
    auto put = [&](uint64_t v, int n) {
        for (int i = 0; i < n; i++) h.push_back((uint8_t)(v >> (8 * i)));
    };
    struct Ent { unsigned tag, type; uint64_t count, value; };
    const int nch = im.nch;
    std::vector<Ent> e = {{256, 4, 1, (uint64_t)im.nx}, {257, 4, 1, (uint64_t)im.ny}, {258, 3, (uint64_t)nch, 32},
                          {259, 3, 1, 1}, {262, 3, 1, (uint64_t)(nch == 3 || nch == 4 ? 2 : 1)}, {273, (unsigned)(big ? 16 : 4), 1, 0},
                          {277, 3, 1, (uint64_t)nch}, {278, 4, 1, (uint64_t)im.ny}, {279, (unsigned)(big ? 16 : 4), 1, nbytes},
                          {284, 3, 1, 1}, {339, 3, (uint64_t)nch, 3}};
    put(0x4949, 2);
    const size_t inl = big ? 8 : 4;
    if (big) { put(43, 2); put(8, 2); put(0, 2); put(16, 8); put(e.size(), 8); }
    else { put(42, 2); put(8, 4); put(e.size(), 2); }
    const size_t ifd_end = h.size() + e.size() * (big ? 20 : 12) + inl;
    // out-of-line arrays (BitsPerSample / SampleFormat with more samples than fit inline) follow the IFD
    size_t extra = ifd_end;
    std::vector<std::pair<size_t, Ent>> ool;
    for (auto &x : e)
        if (x.count * 2 > inl && x.type == 3) {
            ool.push_back({extra, x});
            extra += x.count * 2;
        }
    const size_t data_off = (extra + 15) & ~(size_t)15;
    for (auto &x : e) {
        put(x.tag, 2);
        put(x.type, 2);
        put(x.count, (int)inl);
        uint64_t val = x.tag == 273 ? data_off : x.value;
        bool isool = false;
        for (auto &o : ool)
            if (o.second.tag == x.tag) { val = o.first; isool = true; }
        if (!isool && x.type == 3 && x.count > 1) {  // short arrays that fit inline
            uint64_t packed = 0;
            for (uint64_t i = 0; i < x.count; i++) packed |= (x.value & 0xffff) << (16 * i);
            val = packed;
        }
        put(val, (int)inl);
    }
    put(0, (int)inl);  // no further IFD
    for (auto &o : ool)
        for (uint64_t i = 0; i < o.second.count; i++) put(o.second.value, 2);
    h.resize(data_off, 0);
    bool ok = fwrite(h.data(), 1, h.size(), f) == h.size();
    std::vector<float> row((size_t)im.nx * nch);
    for (int y = 0; y < im.ny && ok; y++) {
        for (int x = 0; x < im.nx; x++)
            for (int c = 0; c < nch; c++) row[(size_t)x * nch + c] = im.data[(size_t)y * im.nx + x + (size_t)c * np];
        ok = fwrite(row.data(), 4, row.size(), f) == row.size();
    }
    ok = fclose(f) == 0 && ok;
    if (!ok) throw std::runtime_error("error writing " + path);
}

}  // namespace tiff

// ------------------------------------------------------------------------------------------------ Netpbm / PFM
namespace pnm {

inline HostImg decode(const bytes &f, const std::string &path)
{
    auto bad = [&](const char *m) { return std::runtime_error(path + ": " + m); };
    size_t pos = 2;
    auto token = [&]() {
        for (;;) {
            while (pos < f.size() && isspace(f[pos])) pos++;
            if (pos < f.size() && f[pos] == '#') {
                while (pos < f.size() && f[pos] != '\n') pos++;
                continue;
            }
            break;
        }
        std::string t;
        while (pos < f.size() && !isspace(f[pos])) t += (char)f[pos++];
        if (t.empty()) throw bad("truncated header");
        return t;
    };
    const char kind = (char)f[1];
    if (kind == 'f' || kind == 'F') {
        // as the reference reads them (iio.c:2561-2578): rows in file order, native byte order
        const int w = atoi(token().c_str()), h = atoi(token().c_str());
        token();  // scale / endianness marker
        pos++;
        const int ch = kind == 'F' ? 3 : 1;
        const size_t n = (size_t)w * h * ch;
        if (w <= 0 || h <= 0 || pos + n * 4 > f.size()) throw bad("truncated PFM");
        const uint8_t *d = &f[pos];
        return planar(w, h, ch, [&](size_t i) { float v; memcpy(&v, d + 4 * i, 4); return v; });
    }
    const int w = atoi(token().c_str()), h = atoi(token().c_str());
    const int ch = (kind == '3' || kind == '6') ? 3 : 1;
    if (w <= 0 || h <= 0) throw bad("bad size");
    const size_t n = (size_t)w * h * ch;
    if (kind == '5' || kind == '6') {
        const int maxv = atoi(token().c_str());
        pos++;
        const int bsz = maxv > 255 ? 2 : 1;
        if (pos + n * bsz > f.size()) throw bad("truncated data");
        const uint8_t *d = &f[pos];
        return planar(w, h, ch, [&](size_t i) { return bsz == 1 ? (float)d[i] : (float)((unsigned)d[2 * i] << 8 | d[2 * i + 1]); });
    }
    if (kind == '2' || kind == '3') {
        token();
        if (n > f.size()) throw bad("truncated data");
        std::vector<float> s(n);
        for (size_t i = 0; i < n; i++) s[i] = (float)atoi(token().c_str());
        return planar(w, h, ch, [&](size_t i) { return s[i]; });
    }
    throw bad("unsupported Netpbm type (bitmaps P1/P4 are not read)");
}

inline void write_pfm(const std::string &path, const HostImg &im)
{
    if (im.nch != 1 && im.nch != 3) throw std::runtime_error(path + ": PFM holds 1 or 3 channels");
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    fprintf(f, "P%c\n%d %d\n-1\n", im.nch == 3 ? 'F' : 'f', im.nx, im.ny);
    const size_t np = (size_t)im.nx * im.ny;
    std::vector<float> inter(np * im.nch);
    for (size_t p = 0; p < np; p++)
        for (int c = 0; c < im.nch; c++) inter[p * im.nch + c] = im.data[p + c * np];
    const bool ok = fwrite(inter.data(), 4, inter.size(), f) == inter.size();
    if (fclose(f) != 0 || !ok) throw std::runtime_error("error writing " + path);
}

}  // namespace pnm

inline bool has_suffix(const std::string &s, const char *suf)
{
    const size_t n = strlen(suf);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; i++)
        if (tolower((unsigned char)s[s.size() - n + i]) != suf[i]) return false;
    return true;
}

// Reads by content, not by name (as iio does).
inline HostImg read(const std::string &path)
{
    const bytes f = slurp(path);
    if (f.size() >= 8 && !memcmp(f.data(), "\x89PNG\r\n\x1a\n", 8)) return png::decode(f, path);
    if (f.size() >= 4 && ((f[0] == 'I' && f[1] == 'I') || (f[0] == 'M' && f[1] == 'M')) &&
        (f[f[0] == 'I' ? 2 : 3] == 42 || f[f[0] == 'I' ? 2 : 3] == 43) && f[f[0] == 'I' ? 3 : 2] == 0)
        return tiff::decode(f, path);
    if (f.size() >= 6 && !memcmp(f.data(), "\x93NUMPY", 6)) return npy::read(path);
    if (f.size() >= 3 && f[0] == 'P' && strchr("2356fF", f[1]) && isspace(f[2])) return pnm::decode(f, path);
    throw std::runtime_error(path + ": unrecognised image format (PNG, TIFF, PGM/PPM, PFM and .npy are read)");
}

// Writes by suffix: float samples to .tif/.tiff, .pfm or .npy.
inline void write(const std::string &path, const HostImg &im)
{
    if (has_suffix(path, ".npy")) return npy::write(path, im);
    if (has_suffix(path, ".tif") || has_suffix(path, ".tiff")) return tiff::write(path, im);
    if (has_suffix(path, ".pfm")) return pnm::write_pfm(path, im);
    throw std::runtime_error(path + ": outputs are float images; use a .tif, .tiff, .pfm or .npy name");
}

}  // namespace imgio
