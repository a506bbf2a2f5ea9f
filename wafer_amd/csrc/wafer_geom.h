// Device layout of one z-slab of the padded grid (DESIGN.md "Data layout in HBM").
//
// The reference keeps Array3<R64> in C-order [x][y][z] (z contiguous).  On the
// device the axes are stored [z][y][x] with x contiguous, so that
//   - a z-slab (the multi-GPU decomposition axis) is one contiguous range and
//     its halo planes need no packing,
//   - wavefront lanes run along x.
// Axis identity is preserved (device x IS the reference's x), only the memory
// order differs; upload/download transpose on the device.
//
// Every row is shifted by `xoff` elements so that the first WORK cell of each
// row (padded x index R) sits on a 128-byte boundary; `pitch` is a multiple of
// 128 bytes.  Frame cells (the Dirichlet zero frame of config.rs:597-622) and
// the pad cells are stored as zeros and are never written by any kernel.
//
// Guard zone: every array is allocated with `gy` extra zero rows on both sides
// of each plane, `gz` extra zero planes on both sides of the slab, and a row
// pitch that covers whole 1 KiB tiles.  Device pointers handed to kernels
// point at (plane 0, row 0), `base_off` elements into the allocation, so a
// tile may read rows / planes / columns just outside the padded grid without a
// bounds predicate -- they exist and hold zeros.  (Per-lane "load or zero"
// selects make hipcc branch around every load and serialise them.)
#pragma once
#include <stdint.h>

struct WaferGeom {
    int nx, ny, nz;     // GLOBAL work-area size (config.grid.size)
    int R;              // CentralDifference::ext()
    int G;              // ghost planes kept on each z side of the slab (>= R)
    int px, py, pzg;    // padded x, y extents and GLOBAL padded z extent (n + 2R)
    int z_begin, nzl;   // owned work planes [z_begin, z_begin + nzl)
    int lz;             // local planes = nzl + 2G
    int xoff;           // element offset of padded x index 0 inside a row
    int pitch;          // row stride in elements
    int gy, gz;         // guard rows per plane side / guard planes per slab side (zeros)
    long long plane;    // plane stride in elements (= (py + 2 gy) * pitch)
    long long total;    // allocation size in elements (= (lz + 2 gz) * plane)
    long long base_off; // element offset of (plane 0, row 0) inside the allocation

    // element offset of (local plane lzp, padded y, padded x)
    __host__ __device__ inline long long at(int lzp, int yp, int xp) const
    {
        return (long long)lzp * plane + (long long)yp * pitch + xoff + xp;
    }
    // global padded z index of local plane lzp (may fall outside [0, pzg))
    __host__ __device__ inline int zp_of(int lzp) const { return z_begin + R + (lzp - G); }
    // local plane of owned work plane kl in [0, nzl)
    __host__ __device__ inline int lzp_of_work(int kl) const { return kl + G; }
};

static inline WaferGeom wafer_make_geom(int nx, int ny, int nz, int R, int G, int z_begin, int nzl,
                                        int elem_bytes)
{
    WaferGeom g;
    g.nx = nx; g.ny = ny; g.nz = nz; g.R = R; g.G = G;
    g.px = nx + 2 * R; g.py = ny + 2 * R; g.pzg = nz + 2 * R;
    g.z_begin = z_begin; g.nzl = nzl; g.lz = nzl + 2 * G;
    const int align = 128 / elem_bytes;           // elements per 128-byte line
    g.xoff = align - R;                           // R <= 3 < align
    const int tile = 8 * align;                   // tile width: 64 lanes x 16 B = 1 KiB of x
    const int nx_tiles = ((nx + tile - 1) / tile) * tile;
    g.pitch = ((g.xoff + R + nx_tiles + 2 * R + align - 1) / align) * align;
    g.gy = 16 + 2 * R + R;                        // tallest tile (16 rows) overhang + 2R halo rows
    g.gz = 3 * R;                                 // the three-step kernel (ext 1) loads up to 3 planes past the slab's ghost planes
                                                  // in either marching direction; the two-step kernel 2R
    g.plane = (long long)(g.py + 2 * g.gy) * g.pitch;
    g.total = (long long)(g.lz + 2 * g.gz) * g.plane;
    g.base_off = (long long)g.gz * g.plane + (long long)g.gy * g.pitch;
    return g;
}
