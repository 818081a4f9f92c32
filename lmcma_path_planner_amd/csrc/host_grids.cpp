// Synthetic cost-map generators (host side), declared in include/lmcma_opt.h.
// S/ = sample_based_optimisation_based_path_planner/src/sample_based_optimisation_based_path_planner.cpp
#include <math.h>
#include <string.h>

#include <algorithm>
#include <limits>
#include <cstdio>
#include <cstring>
#include <vector>

#include "lmcma_opt.h"

namespace {

struct Box { int r0, r1, c0, c1; };

inline uint64_t mix64(uint64_t &state)
{
    // SplitMix64
    uint64_t z = (state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// One pass of the lower-envelope distance transform along a line
// (Felzenszwalb & Huttenlocher): out[q] = min_p (q - p)^2 + f[p].
struct Envelope {
    std::vector<int> site;
    std::vector<double> edge, f, out;
    explicit Envelope(int n) : site(n), edge(n + 1), f(n), out(n) {}
    void run(int n)
    {
        int k = 0;
        site[0] = 0;
        edge[0] = -1e300;
        edge[1] = 1e300;
        for (int q = 1; q < n; ++q) {
            double x;
            while (true) {
                const int p = site[k];
                x = ((f[q] + (double)q * q) - (f[p] + (double)p * p)) / (2.0 * q - 2.0 * p);
                if (x <= edge[k]) --k; else break;
            }
            ++k;
            site[k] = q;
            edge[k] = x;
            edge[k + 1] = 1e300;
        }
        k = 0;
        for (int q = 0; q < n; ++q) {
            while (edge[k + 1] < q) ++k;
            const double dq = (double)q - site[k];
            out[q] = dq * dq + f[site[k]];
        }
    }
};

struct Offset { int dx, dy; };
inline int norm2(Offset o) { return o.dx * o.dx + o.dy * o.dy; }

// sequential 8-neighbour sweep, two passes (S/:447-490)
void sweep8(std::vector<Offset> &g, int w, int h)
{
    const Offset far_away = {9999, 9999};
    auto at = [&](int x, int y) -> Offset {
        return (x >= 0 && y >= 0 && x < w && y < h) ? g[(size_t)y * w + x] : far_away;
    };
    auto relax = [&](Offset &p, int x, int y, int ox, int oy) {
        Offset o = at(x + ox, y + oy);
        o.dx += ox;
        o.dy += oy;
        if (norm2(o) < norm2(p)) p = o;
    };
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            Offset p = at(x, y);
            relax(p, x, y, -1, 0);
            relax(p, x, y, 0, -1);
            relax(p, x, y, -1, -1);
            relax(p, x, y, 1, -1);
            g[(size_t)y * w + x] = p;
        }
        for (int x = w - 1; x >= 0; --x) {
            Offset p = at(x, y);
            relax(p, x, y, 1, 0);
            g[(size_t)y * w + x] = p;
        }
    }
    for (int y = h - 1; y >= 0; --y) {
        for (int x = w - 1; x >= 0; --x) {
            Offset p = at(x, y);
            relax(p, x, y, 1, 0);
            relax(p, x, y, 0, 1);
            relax(p, x, y, -1, 1);
            relax(p, x, y, 1, 1);
            g[(size_t)y * w + x] = p;
        }
        for (int x = 0; x < w; ++x) {
            Offset p = at(x, y);
            relax(p, x, y, -1, 0);
            g[(size_t)y * w + x] = p;
        }
    }
}

}  // namespace

extern "C" {

int lmcma_grid_two_bar(int size, int nz, uint8_t *occ)
{
    if (size < 1 || nz < 1 || !occ) return -1;
    // S/:269-297 marks matrix cells (a, b) with a < 60, 65 <= b < 75 and
    // 40 <= a < 99, 35 <= b < 45; the matrix is read as (row = y, col = x)
    // (S/:599-602), so a is the row.  Extents scale with size / 100.
    const Box bars[2] = {{0, 60 * size / 100, 65 * size / 100, 75 * size / 100},
                         {40 * size / 100, 99 * size / 100, 35 * size / 100, 45 * size / 100}};
    const size_t layer = (size_t)size * size;
    std::fill(occ, occ + layer * nz, (uint8_t)0);
    for (int z = 0; z < nz; ++z) {
        // in 3-D the first bar spans the lower three quarters of z, the second the upper three
        const bool use[2] = {nz == 1 || z < (3 * nz) / 4, nz == 1 || z >= nz / 4};
        for (int b = 0; b < 2; ++b) {
            if (!use[b]) continue;
            for (int r = bars[b].r0; r < bars[b].r1; ++r)
                std::fill(occ + z * layer + (size_t)r * size + bars[b].c0,
                          occ + z * layer + (size_t)r * size + bars[b].c1, (uint8_t)1);
        }
    }
    return 0;
}

int lmcma_grid_random_boxes(int nx, int ny, int nz, uint64_t seed, double fill, const double *start,
                            const double *goal, uint8_t *occ)
{
    if (nx < 1 || ny < 1 || nz < 1 || !occ || !start || !goal) return -1;
    const long cells = (long)nx * ny * nz;
    long filled = std::count_if(occ, occ + cells, [](uint8_t v) { return v != 0; });
    const double keep = 0.05 * nx;  // free radius around the endpoints
    uint64_t st = seed;
    for (int tries = 0; (double)filled < fill * (double)cells && tries < 100000; ++tries) {
        const int bw = 2 + (int)(mix64(st) % (uint64_t)(nx / 16 + 1));
        const int bh = 2 + (int)(mix64(st) % (uint64_t)(ny / 16 + 1));
        const int bd = nz > 1 ? 2 + (int)(mix64(st) % (uint64_t)(nz / 16 + 1)) : 1;
        const int x0 = (int)(mix64(st) % (uint64_t)nx);
        const int y0 = (int)(mix64(st) % (uint64_t)ny);
        const int z0 = nz > 1 ? (int)(mix64(st) % (uint64_t)nz) : 0;
        const int x1 = std::min(x0 + bw, nx), y1 = std::min(y0 + bh, ny), z1 = std::min(z0 + bd, nz);
        bool near_endpoint = false;
        for (const double *q : {start, goal}) {
            const double cx = std::min(std::max(q[0], (double)x0), (double)(x1 - 1));
            const double cy = std::min(std::max(q[1], (double)y0), (double)(y1 - 1));
            const double cz = nz > 1 ? std::min(std::max(q[2], (double)z0), (double)(z1 - 1)) : 0.0;
            const double dz = nz > 1 ? q[2] - cz : 0.0;
            const double d2 = (q[0] - cx) * (q[0] - cx) + (q[1] - cy) * (q[1] - cy) + dz * dz;
            if (d2 < keep * keep) { near_endpoint = true; break; }
        }
        if (near_endpoint) continue;
        for (int z = z0; z < z1; ++z)
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) {
                    uint8_t &c = occ[((size_t)z * ny + y) * nx + x];
                    if (!c) { c = 1; ++filled; }
                }
    }
    return 0;
}

int lmcma_grid_edt(const uint8_t *occ, int nx, int ny, int nz, float *dist)
{
    if (nx < 1 || ny < 1 || nz < 1 || !occ || !dist) return -1;
    const size_t cells = (size_t)nx * ny * nz;
    std::vector<double> g(cells);
    for (size_t i = 0; i < cells; ++i) g[i] = occ[i] ? 0.0 : 1e12;
    Envelope env(std::max(nx, std::max(ny, nz)));
    for (int z = 0; z < nz; ++z)
        for (int y = 0; y < ny; ++y) {
            double *line = &g[((size_t)z * ny + y) * nx];
            std::copy(line, line + nx, env.f.begin());
            env.run(nx);
            std::copy(env.out.begin(), env.out.begin() + nx, line);
        }
    for (int z = 0; z < nz; ++z)
        for (int x = 0; x < nx; ++x) {
            for (int y = 0; y < ny; ++y) env.f[y] = g[((size_t)z * ny + y) * nx + x];
            env.run(ny);
            for (int y = 0; y < ny; ++y) g[((size_t)z * ny + y) * nx + x] = env.out[y];
        }
    if (nz > 1)
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x) {
                for (int z = 0; z < nz; ++z) env.f[z] = g[((size_t)z * ny + y) * nx + x];
                env.run(nz);
                for (int z = 0; z < nz; ++z) g[((size_t)z * ny + y) * nx + x] = env.out[z];
            }
    for (size_t i = 0; i < cells; ++i) dist[i] = (float)sqrt(g[i]);
    return 0;
}

int lmcma_grid_8ssedt(const uint8_t *inside, int width, int height, int32_t *signed_dist,
                      double *dist_to_inside)
{
    if (!inside || width < 1 || height < 1) return -1;
    const size_t n = (size_t)width * height;
    const Offset zero = {0, 0}, far_away = {9999, 9999};
    // grid A: marked cells are the sources; grid B: unmarked cells are (S/:515-523)
    std::vector<Offset> a(n), b(n);
    for (size_t i = 0; i < n; ++i) {
        a[i] = inside[i] ? zero : far_away;
        b[i] = inside[i] ? far_away : zero;
    }
    sweep8(a, width, height);
    sweep8(b, width, height);
    for (size_t i = 0; i < n; ++i) {
        const double da = sqrt((double)norm2(a[i])), db = sqrt((double)norm2(b[i]));
        if (signed_dist) signed_dist[i] = (int)da - (int)db;  // S/:538-540
        if (dist_to_inside) dist_to_inside[i] = da;
    }
    return 0;
}

int lmcma_grid_state_cost(const float *dist, long cells, double d_min, float *cost)
{
    if (!dist || !cost || cells < 1 || !(d_min > 0.0)) return -1;
    // 1 / clearance (S/:667); the reference leaves zero clearance to its validity
    // check (S/:602), here the clearance is floored at d_min instead
    for (long i = 0; i < cells; ++i) cost[i] = (float)(1.0 / std::max((double)dist[i], d_min));
    return 0;
}

// ---- map files ------------------------------------------------------------------------

// binvox: text header ("#binvox 1", "dim d h w", "translate", "scale", "data") then
// (value, count) byte pairs; voxel i is x = i / (w h), z = (i / w) % h, y = i % w
// (binvox2bt.cpp:164-255 of the reference's planner package)
int lmcma_grid_read_binvox(const char *path, int dims[3], double translate[3], double *scale, uint8_t *occ,
                           long capacity)
{
    if (!path || !dims) return -1;
    FILE *f = fopen(path, "rb");
    if (!f) return -2;
    struct Closer { FILE *f; ~Closer() { fclose(f); } } closer{f};
    char word[64];
    int version = 0;
    if (fscanf(f, "%63s %d", word, &version) != 2 || strcmp(word, "#binvox") != 0) return -3;
    int depth = -1, height = -1, width = -1;
    double t[3] = {0, 0, 0}, sc = 1.0;
    bool data = false;
    while (!data && fscanf(f, "%63s", word) == 1) {
        if (!strcmp(word, "data")) data = true;
        else if (!strcmp(word, "dim")) { if (fscanf(f, "%d %d %d", &depth, &height, &width) != 3) return -3; }
        else if (!strcmp(word, "translate")) { if (fscanf(f, "%lf %lf %lf", &t[0], &t[1], &t[2]) != 3) return -3; }
        else if (!strcmp(word, "scale")) { if (fscanf(f, "%lf", &sc) != 1) return -3; }
        else { int ch; while ((ch = fgetc(f)) != EOF && ch != '\n') {} }  // unknown keyword: skip its line
    }
    if (!data || depth < 1 || height < 1 || width < 1 || depth > 65536 || height > 65536 || width > 65536) return -3;
    dims[0] = depth; dims[1] = width; dims[2] = height;  // extents along x, y, z
    if (translate) std::copy(t, t + 3, translate);
    if (scale) *scale = sc;
    const long size = (long)depth * height * width;
    if (!occ) return 0;
    if (capacity < size) return -4;
    if (fgetc(f) == EOF) return -3;  // the line feed after "data"
    long index = 0, occupied = 0;
    const int nx = depth, ny = width;
    while (index < size) {
        const int value = fgetc(f), count = fgetc(f);
        if (value == EOF || count == EOF) return -5;  // truncated
        if (index + count > size) return -5;
        for (long i = index; i < index + count; ++i) {
            const long y = i % width, z = (i / width) % height, x = i / ((long)width * height);
            occ[(z * ny + y) * nx + x] = value ? 1 : 0;
        }
        if (value) occupied += count;
        index += count;
    }
    return occupied > 0x7fffffff ? 0x7fffffff : (int)occupied;
}

// Uncompressed BMP (8-bit palette, 24- or 32-bit pixels); a pixel is marked when its green
// channel is below 128, the test of Signed_Distance_Fields_test (S/:507-523).  Rows come
// out top to bottom like the SDL surface the reference walks.
int lmcma_grid_read_bmp(const char *path, int *width, int *height, uint8_t *marked, long capacity)
{
    if (!path || !width || !height) return -1;
    FILE *f = fopen(path, "rb");
    if (!f) return -2;
    struct Closer { FILE *f; ~Closer() { fclose(f); } } closer{f};
    unsigned char h[54];
    if (fread(h, 1, 54, f) != 54 || h[0] != 'B' || h[1] != 'M') return -3;
    auto u32 = [&](int at) { return (uint32_t)h[at] | (uint32_t)h[at + 1] << 8 | (uint32_t)h[at + 2] << 16 | (uint32_t)h[at + 3] << 24; };
    auto u16 = [&](int at) { return (uint32_t)h[at] | (uint32_t)h[at + 1] << 8; };
    const uint32_t offset = u32(10), header = u32(14);
    const int32_t w = (int32_t)u32(18), hh = (int32_t)u32(22);
    const uint32_t bpp = u16(28), compression = u32(30);
    if (header < 40 || w < 1 || hh == 0 || (compression != 0 && compression != 3)) return -3;
    if (bpp != 8 && bpp != 24 && bpp != 32) return -3;
    const int H = hh < 0 ? -hh : hh;
    *width = w;
    *height = H;
    if (!marked) return 0;
    if (capacity < (long)w * H) return -4;
    std::vector<unsigned char> palette;
    if (bpp == 8) {
        uint32_t colours = u32(46);
        if (colours == 0 || colours > 256) colours = 256;
        palette.resize(4 * 256, 0);
        if (fseek(f, 14 + (long)header, SEEK_SET) != 0 || fread(palette.data(), 4, colours, f) != colours) return -5;
    }
    const size_t stride = (((size_t)w * bpp + 31) / 32) * 4;
    std::vector<unsigned char> row(stride);
    if (fseek(f, (long)offset, SEEK_SET) != 0) return -5;
    int count = 0;
    for (int r = 0; r < H; ++r) {
        if (fread(row.data(), 1, stride, f) != stride) return -5;
        const int y = hh < 0 ? r : H - 1 - r;  // positive height: the file starts with the bottom row
        for (int x = 0; x < w; ++x) {
            // pixels are stored blue, green, red(, alpha)
            const unsigned char g = bpp == 8 ? palette[4 * row[x] + 1] : row[(size_t)x * (bpp / 8) + 1];
            const uint8_t m = g < 128 ? 1 : 0;
            marked[(size_t)y * w + x] = m;
            count += m;
        }
    }
    return count;
}

}  // extern "C"
