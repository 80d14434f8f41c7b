/*
 * lsrecon.h -- C ABI of the MI355X (gfx950) light-sheet reconstruction hot path.
 *
 * This is the drop-in boundary: a flat, extern "C" interface with plain pointers and sizes.
 * No torch types, no allocation inside, no global scratch: every buffer is a DEVICE pointer
 * owned by the caller (in the Python host: a torch-ROCm tensor's data_ptr()), every entry
 * point takes the HIP stream to launch on and is re-entrant across streams.
 *
 * What each entry point replaces in the reference (czbiohub-sf/shrimPy, paths relative to the
 * reference root; biahub = un-vendored dependency pinned at pyproject.toml:91):
 *
 *   lsr_deskew_f32          biahub.deskew.fast_deskew_zyx, called at
 *                           shrimpy/preprocessing.py:408-413 (and the older
 *                           biahub deskew_data, scripts/measure_psf.py:238-246)
 *   lsr_affine_f32          registration apply (north-star; docs/data_structure.md:58-62) ==
 *                           scipy.ndimage.affine_transform(order=1); also the general-matrix
 *                           deskew path together with lsr_average_slices_f32
 *   lsr_average_slices_f32  biahub _average_n_slices (average_n_slices setting,
 *                           config/mda/mantis/dynatrack_demo.yaml:164)
 *   lsr_correlate_*, lsr_rl_*  Richardson-Lucy deconvolution (north-star; no reference symbol)
 *
 * Layout: all volumes are C-order (Z, Y, X) float32, X fastest, densely packed.
 * Matrices: `M` is a row-major 3x4 double, OUTPUT index -> INPUT coordinate (the
 * scipy.ndimage convention): in_c = M[c][0]*zo + M[c][1]*yo + M[c][2]*xo + M[c][3].
 *
 * Return value: 0 = ok; < 0 = argument error (LSR_E_*), nothing was launched;
 * > 0 = a hipError_t from the launch. lsr_last_error() returns a thread-local message.
 * Nothing here synchronises the stream or the device.
 *
 * Sizes: every extent in [1, 2^30), fewer than 2^48 voxels per volume, row strides below 2^31 and plane strides
 * below 2^32 elements -- anything else is LSR_E_UNSUPPORTED before the entry computes with it.  Inside those bounds
 * volumes of more than 2^32 voxels are ordinary (64-bit bases per plane / row); narrower limits are stated at the entries.
 */
#ifndef LSRECON_H
#define LSRECON_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSR_VERSION 100 /* 0.1.0 */

#define LSR_OK 0
#define LSR_E_NULL (-1)        /* a required pointer is NULL */
#define LSR_E_SHAPE (-2)       /* a dimension is <= 0 or inconsistent */
#define LSR_E_UNSUPPORTED (-3) /* valid request this kernel does not cover (see each function) */
#define LSR_E_ARG (-4)         /* another argument is out of range */

/* Border rule of the order-1 sampler. */
#define LSR_MODE_CONSTANT 0      /* scipy mode="constant": any coordinate outside [0,n-1] -> cval */
#define LSR_MODE_GRID_CONSTANT 1 /* scipy mode="grid-constant": blend towards cval over one voxel */
/* OR-able flag for lsr_affine_f32: interpolate in f32 (coordinates and border decisions stay
 * fp64). Not bit-identical to scipy (~1e-6 relative) but HBM-bound instead of fp64-VALU-bound. */
#define LSR_MODE_F32_INTERP 256

/* Which half of a Richardson-Lucy iteration a correlation launch finishes (fused epilogue). */
#define LSR_EPI_NONE 0   /* out = corr(in)                                  (plain correlation) */
#define LSR_EPI_RATIO 1  /* out = aux / (corr(in) + eps)       aux = y      (ratio = y/(Hx+eps)) */
#define LSR_EPI_UPDATE 2 /* out = aux * corr(in) / norm        aux = x      (x <- x*H^T(ratio)/H^T1) */
#define LSR_EPI_SCALE 3  /* out = corr(in) / norm   (lsr_correlate_dense_padded_f32 only: the (z, x) half
                            of a PSF that separates along y, whose y half runs as a separable pass) */

typedef void* lsr_stream_t; /* a hipStream_t; NULL = the default stream */

int lsr_version(void);
const char* lsr_last_error(void);
/* Fingerprint of the sources this BINARY was compiled from: the first 16 hex digits of the sha256 over (name, bytes) of
 * include/lsrecon.h, csrc/Makefile and every .hip / .hpp file in csrc (csrc/Makefile computes it, api.o is rebuilt whenever
 * one of them changes).  The library is git-ignored and travels to the GPU box as a file: the Python host compares this
 * with the same hash of the checkout beside it (shrimpy_amd/_lib.py::kernel_source_sha16) and refuses a stale build. */
const char* lsr_source_sha16(void);
/* Page-locked host memory of exactly `bytes` bytes from the HIP runtime (hipHostMalloc, portable across
 * devices): the staging slots of the store-to-store path.  Returns 0 or the hipError_t (then *out = NULL). */
int lsr_pinned_alloc(int64_t bytes, void** out);
int lsr_pinned_free(void* ptr);

/*
 * Oblique-plane deskew with the slice averaging fused in.
 *
 *   in   raw stack (Z, Y, X) = (scan, tilt, coverslip)
 *   out  (Zo, Yo, Xo) with strides out_pitch (y) and out_plane (z) in floats -- dense: Xo and
 *        Yo*Xo; a padded volume (lsr_sep_padded_shape) lets the deskew write the RL input in
 *        place, on cache-line-aligned rows; Zo == ceil(Zd / avg_n), Zd = depth BEFORE averaging
 *   M    output->input map over the PRE-average grid (Zd, Yo, Xo). It must have the deskew
 *        structure: row 0 = (a, 0, b, c) (only z_in is interpolated), row 1 = (+-1, 0, 0, int),
 *        row 2 = (0, +-1, 0, int). Anything else returns LSR_E_UNSUPPORTED: use
 *        lsr_affine_f32 + lsr_average_slices_f32 for a general matrix.
 *   avg_n  >= 1; pre-average slices zd = zo*avg_n + k, k < avg_n, clamped to Zd-1 (edge pad).
 *
 * Arithmetic: coordinates and the 2-tap interpolation in fp64 exactly as
 * scipy.ndimage.affine_transform(order=1, mode="constant", cval=0) evaluates them (result
 * rounded to f32 once), then ((d0+d1)+...)/avg_n in f32: bit-identical to the CPU oracle.
 */
int lsr_deskew_f32(const float* in, int64_t Z, int64_t Y, int64_t X, float* out, int64_t Zo,
                   int64_t Yo, int64_t Xo, int64_t out_pitch, int64_t out_plane, int64_t Zd,
                   const double M[12], int avg_n, lsr_stream_t stream);
/* The same with the raw stack as the camera's uint16 counts: the conversion to float32 (exact)
 * happens on the way into the kernel, so neither the host nor HBM ever holds a float copy of the
 * stack -- half the PCIe upload, half the HBM read. Results equal lsr_deskew_f32 on the converted
 * stack bit for bit. */
int lsr_deskew_u16(const uint16_t* in, int64_t Z, int64_t Y, int64_t X, float* out, int64_t Zo,
                   int64_t Yo, int64_t Xo, int64_t out_pitch, int64_t out_plane, int64_t Zd,
                   const double M[12], int avg_n, lsr_stream_t stream);

/*
 * The same kernel under an explicit border rule, all input variants behind one entry: in_u16 != 0 reads
 * uint16 counts, flat_pattern / flat_mean (both or neither NULL) fuse the flat-field correction, and
 * mode = LSR_MODE_CONSTANT | LSR_MODE_GRID_CONSTANT.  "grid-constant" continues the raw stack with zeros, so
 * a scan coordinate within one sample of either end blends its inside neighbour with 0 -- what
 * scipy mode="grid-constant" and torch grid_sample(padding_mode="zeros") do; the deskew geometry is
 * [RECALLED], this is the switch that covers the other border convention at full
 * speed.  Bit-identical to lsr_affine_f32(mode) followed by lsr_average_slices_f32.
 */
int lsr_deskew_border(const void* in, int in_u16, int64_t Z, int64_t Y, int64_t X, float* out, int64_t Zo,
                      int64_t Yo, int64_t Xo, int64_t out_pitch, int64_t out_plane, int64_t Zd,
                      const double M[12], int avg_n, int mode, const float* flat_pattern,
                      const float* flat_mean, lsr_stream_t stream);
/* ... with the value outside the stack (scipy's `cval`; [RECALLED] biahub's deskew_data takes one and fills with the
 * stack's minimum when it is None): `cval` is a DEVICE scalar -- a constant the caller uploaded, or out2[0] of
 * lsr_minmax_f32 / lsr_minmax_u16 over the stack, so that "min" needs no host round trip -- NULL = 0 (the entries above).
 * Under "grid-constant" the outside neighbour is cval * weight, as scipy sums it.  The _cpu twin reads a host scalar and
 * has the "constant" border only. */
int lsr_deskew_cval(const void* in, int in_u16, int64_t Z, int64_t Y, int64_t X, float* out, int64_t Zo, int64_t Yo,
                    int64_t Xo, int64_t out_pitch, int64_t out_plane, int64_t Zd, const double M[12], int avg_n, int mode,
                    const float* flat_pattern, const float* flat_mean, const float* cval, lsr_stream_t stream);
int lsr_deskew_cval_cpu(const void* in, int in_u16, int64_t Z, int64_t Y, int64_t X, float* out, int64_t Zo, int64_t Yo,
                        int64_t Xo, int64_t out_pitch, int64_t out_plane, int64_t Zd, const double M[12], int avg_n,
                        int mode, const float* flat_pattern, const float* flat_mean, const float* cval, lsr_stream_t stream);
/* min / max of uint16 camera counts as two floats (exact), scratch as lsr_minmax_f32 */
int lsr_minmax_u16(const uint16_t* in, int64_t n, float* out2, void* scratch, lsr_stream_t stream);

/*
 * General order-1 (trilinear) affine resample; any 3x4 matrix.
 * mode/cval as scipy.ndimage.affine_transform. fp64 coordinates and weights, same operation
 * order as scipy: bit-identical to the CPU oracle for finite inputs.
 */
int lsr_affine_f32(const float* in, int64_t Zi, int64_t Yi, int64_t Xi, float* out, int64_t Zo,
                   int64_t Yo, int64_t Xo, const double M[12], float cval, int mode,
                   lsr_stream_t stream);
/*
 * The same with a source whose rows are padded: in_pitch floats from row to row, in_plane from plane to
 * plane.  The LDS-staged kernels move 16-byte chunks, so they need rows that start on 16-byte
 * boundaries: in_pitch and in_plane multiples of 4, in_pitch >= Xi rounded up to 4, `in` 16-byte aligned,
 * and FINITE values in the padding columns [Xi, in_pitch) (they only ever meet weight 0; zeros are the
 * natural choice).  A deskewed volume is Xp = ceil(Z / r - Y cos(theta)) wide -- a multiple of 4 one time
 * in four; lsr_deskew_* writes any out_pitch, so the deskew -> register chain keeps the fast kernels for
 * every width.  Other strides run the gather kernel, as lsr_affine_f32 does for Xi % 4 != 0.
 * out_pitch / out_plane (any values >= the dense ones) let the result land inside a larger allocation, e.g.
 * the padded volume the Richardson-Lucy kernels read, so no copy sits between registration and deconvolution.
 */
int lsr_affine_pitched_f32(const float* in, int64_t Zi, int64_t Yi, int64_t Xi, int64_t in_pitch,
                           int64_t in_plane, float* out, int64_t Zo, int64_t Yo, int64_t Xo,
                           int64_t out_pitch, int64_t out_plane, const double M[12], float cval, int mode,
                           lsr_stream_t stream);
int lsr_affine_path_pitched(int64_t Zi, int64_t Yi, int64_t Xi, int64_t in_pitch, int64_t in_plane,
                            const double M[12], int mode); /* lsr_affine_path for such a source */
/* Which kernel lsr_affine_f32 runs for this matrix on a (.., Yi, Xi) moving volume: 1 = the
 * LDS-staged z-marching kernel (csrc/affine_planar.hip: constant mode, z decoupled from the plane,
 * |M[0]| <= 1.5, Xi a multiple of 4, source box of a 32 x 128 tile within LDS), 0 = the general
 * gather kernel. Results are identical either way. */
int lsr_affine_kernel_choice(int64_t Yi, int64_t Xi, const double M[12], int mode);
/* The same question with all three kernels: 1 = csrc/affine_planar.hip (as above); 2 =
 * csrc/affine_box.hip (constant mode, any matrix -- maps that couple z with the plane included --
 * whose source box of an 8 x 16 x 64 output block fits in 150 KB of LDS, Xi a multiple of 4);
 * 0 = the general gather kernel. Results are identical whichever runs. */
int lsr_affine_path(int64_t Zi, int64_t Yi, int64_t Xi, const double M[12], int mode);
/* Host-side geometry of the box kernel for this matrix: out6 = {block z, y, x extents; staged source
 * box planes, rows, floats per row}; returns 1, or 0 when the box kernel does not apply.  (What the
 * CPU tests check the kernel's coverage claim against: every tap of every voxel of a block lies inside
 * the box spanned from the block's two extreme corners.) */
int lsr_affine_box_shape(int64_t Zi, int64_t Yi, int64_t Xi, const double M[12], int* out6);

/*
 * Affine registration, estimate half (no reference symbol, docs/data_structure.md:58-62).
 * Normal equations of one Gauss-Newton step of   min sum_x (gain * moving(M x) + offset - target(x))^2
 * over the target grid sampled every `stride[axis]` voxels (trilinear taps; a sample counts when all eight
 * lie inside `moving`).  Parameters: the 3x4 matrix row by row in centred, scaled target coordinates
 * ((x - centre) / scale, 1), then gain, offset (14).  `partial` receives lsr_affine_normal_blocks()
 * rows of lsr_affine_normal_size() doubles (105 upper-triangle entries of J^T J row by row, 14 of
 * J^T r, sum r^2, sample count); their sum over rows, in row order, is the result.
 */
int lsr_affine_normal_size(void);
int lsr_affine_normal_blocks(void);
int lsr_affine_normal_equations_f32(const float* moving, int64_t Zi, int64_t Yi, int64_t Xi,
                                    const float* target, int64_t Zo, int64_t Yo, int64_t Xo,
                                    const double M[12], double gain, double offset, const int stride[3],
                                    const double centre[3], double scale, double* partial,
                                    lsr_stream_t stream);

/*
 * Ingest, host side (no device work): the acquisition writes blosc frames (zstd, byte
 * shuffle; shrimpy/mantis/mantis_engine.py:474-481) and the reference reads them through iohub -> zarr ->
 * numcodecs.blosc.decompress.  lsr_blosc_decode_host walks one c-blosc 1.x frame, entropy-decodes every stream
 * and undoes the byte shuffle into `out` (out_bytes = the frame's nbytes; host memory, e.g. a slab of a pinned
 * staging slot).  One call per chunk, safe from many threads at once (a Python walk costs ~10 us per stream
 * under the GIL; c-blosc frames have ~1000 streams per chunk).  *typesize (may be NULL) = the frame's element
 * size.  zstd / lz4 / zlib come from the system libraries at run time (dlopen).
 * LSR_E_UNSUPPORTED: bit shuffle, blosclz / snappy, or the decoder library is missing (callers fall back to
 * another codec); LSR_E_ARG: corrupt frame; LSR_E_SHAPE: out_bytes differs from the frame's nbytes.
 */
int lsr_blosc_host_codec(int compressor); /* 1 when the decoder of blosc compressor code 1 (lz4) / 3 (zlib) / 4 (zstd) is loadable */
int lsr_blosc_decode_host(const uint8_t* frame, int64_t frame_bytes, uint8_t* out, int64_t out_bytes, int* typesize);
/*
 * The writer's side (host code): one c-blosc 1.x frame of `src` with zstd streams, one per block (not split), byte
 * shuffle (shuffle = 1, typesize > 1) or none (0) -- the format the acquisition engine writes and the CLI's default
 * output (mantis_engine.py:474-481).  Byte for byte the frames of the Python encoder (shrimpy_amd/io/codecs.py:
 * blocksize or 256 KB cut to whole elements, a block zstd does not shrink stored verbatim, a frame that does not shrink
 * in the "memcpyed" form), but callable from many threads at once.  dst: lsr_blosc_encode_bound(nbytes, typesize,
 * blocksize) bytes; *out_bytes = the frame's size.  LSR_E_UNSUPPORTED: libzstd's compressor is not loadable, or a
 * shuffle other than 0 / 1 (callers keep the Python encoder).
 */
int lsr_blosc_host_encoder(void);
int64_t lsr_blosc_encode_bound(int64_t nbytes, int typesize, int64_t blocksize);
int lsr_blosc_encode_host(const uint8_t* src, int64_t nbytes, int typesize, int clevel, int shuffle, int64_t blocksize,
                          uint8_t* dst, int64_t cap, int64_t* out_bytes);
/*
 * The writer's side ON THE DEVICE (csrc/blosc_encode.hip): the volume at `src` (src_bytes of `typesize`-byte elements,
 * 1 / 2 / 4) cut into consecutive chunks of frame_bytes -- the Zarr chunks of whole z planes -- each written as one
 * c-blosc 1.x frame (zstd, byte shuffle, blocks of `blocksize` bytes, 0 = 64 K elements; at most 64 K elements per
 * block), the last chunk zero-padded to frame_bytes as Zarr pads edge chunks.  Every shuffled byte plane of a block is
 * one zstd block: RLE, Huffman-coded literals (four streams, no sequences) or Raw -- a subset of the format that any
 * zstd decoder reads and that matches zstd level 1 (the acquisition's setting, mantis_engine.py:474-481) on shuffled
 * image data to ~1 %.  Replaces numcodecs' Blosc(cname="zstd").encode behind iohub's writer
 * (shrimpy/dynatrack/tracking.py:1337-1367) for results that are already in HBM.
 * lsr_blosc_encode_device_plan: the frame count, the scratch bytes and the capacity of `out` the call needs.
 * On return (stream order) frames[2 f] = byte offset of frame f in `out` (16-byte aligned), frames[2 f + 1] = its size.
 * scratch / out / frames: device memory, 16-byte aligned.  The _cpu twin takes host pointers and writes the same bytes.
 */
int lsr_blosc_encode_device_plan(int64_t src_bytes, int typesize, int64_t frame_bytes, int64_t blocksize,
                                 int64_t* n_frames, int64_t* scratch_bytes, int64_t* out_cap);
int lsr_blosc_encode_device(const void* src, int64_t src_bytes, int typesize, int64_t frame_bytes, int64_t blocksize,
                            void* scratch, int64_t scratch_bytes, uint8_t* out, int64_t out_cap, int64_t* frames,
                            lsr_stream_t stream);
int lsr_blosc_encode_device_cpu(const void* src, int64_t src_bytes, int typesize, int64_t frame_bytes, int64_t blocksize,
                                void* scratch, int64_t scratch_bytes, uint8_t* out, int64_t out_cap, int64_t* frames,
                                lsr_stream_t stream);
/*
 * The reader's side ON THE DEVICE (csrc/blosc_decode.hip, csrc/zstd_lane.hpp): `comp` holds the compressed chunks of
 * one volume as the reader threads pread() them from the shard (device memory, comp_bytes long); frames[2 f], frames[2 f + 1]
 * = byte offset and size of chunk f's c-blosc frame in it (size 0: an absent chunk, zero-filled).  Every frame must
 * decode to frame_nbytes bytes in blocks of `blocksize` with elements of `typesize` (1 / 2 / 4) -- the values of the first
 * frame's header, which the host reads before the upload -- and use the zstd compressor (byte shuffle or none); chunk f
 * lands at out + f * frame_nbytes, the last one cut at out_bytes.  One lane per blosc block runs a complete zstd decoder
 * (RFC 8878: Huffman / FSE literals, sequences, repeat offsets), a second launch undoes the shuffle.  Replaces
 * numcodecs' Blosc.decode behind iohub's reader (shrimpy/replay_camera.py:176-268) for stacks on their way into HBM.
 * *status (device, 8 bytes) is 0 afterwards, or (block + 1) << 8 | code of the first block that failed (1 corrupt,
 * 2 destination too small, 3 unsupported, 4 size mismatch) -- a damaged chunk never faults.  The _cpu twin runs the same
 * decoder from host pointers.  lsr_zstd_lane_decode_cpu: one bare zstd frame through that decoder (tests, fuzzing).
 */
int lsr_blosc_decode_device_plan(int64_t n_frames, int64_t frame_nbytes, int64_t blocksize, int typesize,
                                 int64_t* scratch_bytes);
int lsr_blosc_decode_device(const uint8_t* comp, int64_t comp_bytes, const int64_t* frames, int64_t n_frames,
                            int64_t frame_nbytes, int64_t blocksize, int typesize, uint8_t* out, int64_t out_bytes,
                            void* scratch, int64_t scratch_bytes, unsigned long long* status, lsr_stream_t stream);
int lsr_blosc_decode_device_cpu(const uint8_t* comp, int64_t comp_bytes, const int64_t* frames, int64_t n_frames,
                                int64_t frame_nbytes, int64_t blocksize, int typesize, uint8_t* out, int64_t out_bytes,
                                void* scratch, int64_t scratch_bytes, unsigned long long* status, lsr_stream_t stream);
int lsr_zstd_lane_decode_cpu(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap, int64_t* out_n);
/* CRC-32C (Castagnoli, the Zarr v3 `crc32c` codec: shard index, optionally every chunk) of n host bytes; seed = 0, or
 * the value of the bytes before `data` when a buffer is checked in pieces.  SSE4.2 crc32 instruction where the CPU has
 * it, slice-by-8 tables otherwise (lsr_crc32c_host_portable: always the tables -- the cross-check). */
int lsr_crc32c_host(const uint8_t* data, int64_t n, uint32_t seed, uint32_t* out);
int lsr_crc32c_host_portable(const uint8_t* data, int64_t n, uint32_t seed, uint32_t* out);

/* out[zo] = mean_k in[min(zo*avg_n + k, Zd-1)], k < avg_n, f32, ((d0+d1)+...)/avg_n. */
int lsr_average_slices_f32(const float* in, int64_t Zd, int64_t Y, int64_t X, float* out,
                           int64_t Zo, int avg_n, lsr_stream_t stream);

/*
 * Bright-field flat-field correction -- replaces _LabelfreePreprocessor._flat_field_BF
 * (shrimpy/preprocessing.py:385-404): pattern = median over Z per (y, x) pixel with
 * torch.quantile(0.5) semantics (exact middle elements; b - (b - a) * 0.5 between the two of an
 * even count; NaN if the column holds one), out = in / pattern * mean(pattern).
 *
 * lsr_flatfield_pattern_f32: `pattern` (Y*X floats) and `mean_out` (one float) are device
 *   outputs; `scratch` = lsr_flatfield_scratch_bytes() bytes of device memory. Z < 65536.
 *   Reads `in` at most four times (radix select, csrc/flatfield.hip), writes nothing else.
 * lsr_flatfield_apply_f32: out = in / pattern * mean_dev[0] (in place allowed).
 * lsr_deskew_flat_f32: lsr_deskew_f32 with the correction applied to every raw sample on its
 *   way into the kernel -- bit-identical to apply followed by deskew, without the corrected
 *   volume's round trip through HBM.
 */
int lsr_flatfield_scratch_bytes(void);
int lsr_flatfield_pattern_f32(const float* in, int64_t Z, int64_t Y, int64_t X, float* pattern,
                              float* mean_out, void* scratch, lsr_stream_t stream);
int lsr_flatfield_apply_f32(const float* in, const float* pattern, const float* mean_dev,
                            float* out, int64_t Z, int64_t Y, int64_t X, lsr_stream_t stream);
/* The same three for a raw stack of uint16 camera counts (converted exactly inside the kernels;
 * the median then always takes the two-pass select on 16-bit keys). */
int lsr_flatfield_pattern_u16(const uint16_t* in, int64_t Z, int64_t Y, int64_t X, float* pattern,
                              float* mean_out, void* scratch, lsr_stream_t stream);
int lsr_flatfield_apply_u16(const uint16_t* in, const float* pattern, const float* mean_dev,
                            float* out, int64_t Z, int64_t Y, int64_t X, lsr_stream_t stream);
int lsr_deskew_flat_u16(const uint16_t* in, int64_t Z, int64_t Y, int64_t X, float* out, int64_t Zo,
                        int64_t Yo, int64_t Xo, int64_t out_pitch, int64_t out_plane, int64_t Zd,
                        const double M[12], int avg_n, const float* flat_pattern,
                        const float* flat_mean, lsr_stream_t stream);
int lsr_deskew_flat_f32(const float* in, int64_t Z, int64_t Y, int64_t X, float* out, int64_t Zo,
                        int64_t Yo, int64_t Xo, int64_t out_pitch, int64_t out_plane, int64_t Zd,
                        const double M[12], int avg_n, const float* flat_pattern,
                        const float* flat_mean, lsr_stream_t stream);

/*
 * 3-D correlation with zero-padded borders and a fused Richardson-Lucy epilogue:
 *
 *   c[z,y,x] = sum_{a,b,c} w[a,b,c] * in[z+a-pz/2, y+b-py/2, x+c-px/2]      (0 outside)
 *   out      = epilogue(c, aux)                                  (LSR_EPI_*)
 *
 * scipy.ndimage.correlate(in, w, mode="constant") semantics; pass the flipped PSF for
 * H x = ndimage.convolve(x, psf) and the PSF itself for H^T r. pz, py, px odd, <= 15.
 *
 * Dense form: `w` = pz*py*px device floats, C-order.
 * Separable form: `wz`, `wy`, `wx` device float arrays of pz, py, px taps (w = wz x wy x wx).
 *
 * `aux` (epilogues RATIO / UPDATE) and `out` are (Z, Y, X); `out` may alias `aux`, it must
 * not alias `in`. For LSR_EPI_UPDATE the divisor norm = H^T 1 (the sum of the taps that land
 * inside the volume) comes from `nz`, `ny`, `nx`:
 *   separable: device arrays of Z, Y, X floats, norm = nz[z]*ny[y]*nx[x];
 *   dense:     `norm_table` = device doubles, the (pz+1)*(py+1)*(px+1) inclusive 3-D prefix
 *              sum of w with a leading zero plane/row/column:
 *              P[a][b][c] = sum_{a'<a, b'<b, c'<c} w[a'][b'][c'].
 */
int lsr_correlate_sep_f32(const float* in, float* out, const float* aux, int64_t Z, int64_t Y,
                          int64_t X, const float* wz, int pz, const float* wy, int py,
                          const float* wx, int px, int epilogue, float eps, const float* nz,
                          const float* ny, const float* nx, lsr_stream_t stream);

int lsr_correlate_dense_f32(const float* in, float* out, const float* aux, int64_t Z, int64_t Y,
                            int64_t X, const float* w, int pz, int py, int px, int epilogue,
                            float eps, const double* norm_table, lsr_stream_t stream);

/*
 * The tuned (HBM-bound) separable kernel reads its input through a ZERO HALO instead of bounds
 * checks: zero padding is real memory, every load is unconditional and 16-byte aligned.
 *
 * lsr_sep_padded_shape: for a (Y, X) plane and a pz x py x px PSF returns
 *   shape[0] = rows and shape[1] = pitch (floats, multiple of 4) of the padded plane,
 *   shape[2], shape[3] = row / column of the logical element (y=0, x=0) inside it.
 * A padded volume is Z such planes, contiguous, base 128-byte aligned; everything outside the
 * logical (Y, X) window must be zero (the kernels only ever write inside it, so zero it once).
 * The logical origin sits at column 32 so that output runs start on 128-byte lines.
 *
 * lsr_correlate_sep_strided_f32: as lsr_correlate_sep_f32, with explicit strides (floats).
 * `in`, `aux`, `out` point at the LOGICAL element (0,0,0) of their volume; `in` must live in a
 * padded volume as above; `aux` / `out` may be dense (pitch X, plane Y*X) or padded.
 */
int lsr_sep_padded_shape(int64_t Y, int64_t X, int pz, int py, int px, int64_t shape[4]);

int lsr_correlate_sep_strided_f32(const float* in, int64_t in_pitch, int64_t in_plane,
                                  const float* aux, int64_t aux_pitch, int64_t aux_plane,
                                  float* out, int64_t out_pitch, int64_t out_plane, int64_t Z,
                                  int64_t Y, int64_t X, const float* wz, int pz, const float* wy,
                                  int py, const float* wx, int px, int epilogue, float eps,
                                  const float* nz, const float* ny, const float* nx,
                                  lsr_stream_t stream);

/*
 * Whole Richardson-Lucy loop: `iters` x { ratio = y/(H x + eps); x <- x * H^T ratio / H^T 1 }.
 * Launches 2*iters kernels on `stream`; capturable in a hipGraph (no sync, no allocation).
 *
 * Separable: k* = PSF factors along z, y, x, k*_flipped the same reversed (H = correlation with
 * the flipped taps). `x_pad` and `ratio_pad` are padded volumes (lsr_sep_padded_shape) with zero
 * halos: the caller writes the initial estimate into the logical window of `x_pad` -- or sets
 * `init_from_y`, which starts from x = y without that copy (then `y` itself must live in a padded
 * volume with a zero halo, e.g. written there by lsr_deskew_f32); `ratio_pad` is scratch. The final estimate is written to the dense (Z, Y, X) `x_out`, or left
 * in `x_pad` if `x_out` is NULL. `y` points at its logical (0,0,0) with strides y_pitch / y_plane
 * (dense: X and Y*X; a padded y keeps the ratio launch's reads on cache-line boundaries).
 *
 * Dense: psf = pz*py*px floats, psf_flipped = the same reversed on all three axes; x (dense) is
 * updated in place (caller initialises it), `ratio` is dense scratch.
 */
int lsr_rl_sep_f32(const float* y, int64_t y_pitch, int64_t y_plane, int init_from_y,
                   float* x_pad, float* ratio_pad, float* x_out, int64_t Z, int64_t Y, int64_t X,
                   const float* kz, const float* kz_flipped, int pz,
                   const float* ky, const float* ky_flipped, int py, const float* kx,
                   const float* kx_flipped, int px, const float* nz, const float* ny,
                   const float* nx, int iters, float eps, lsr_stream_t stream);

/*
 * The same iterations with ONE launch each (rl_fused_sep.hip): ratio = y / (H x + eps) is formed
 * and consumed inside the workgroup, so an iteration moves 12 bytes per voxel through HBM instead
 * of 24. Results are bit-identical to lsr_rl_sep_f32. Compiled for every odd tap count up to 15 per
 * axis except 15 z taps with 11+ in-plane taps (lsr_rl_sep_fused_supported: 1 / 0; otherwise
 * LSR_E_UNSUPPORTED: use lsr_rl_sep_f32).
 *
 * `y` points at the logical (0,0,0) of a padded volume (lsr_sep_padded_shape geometry or larger
 * strides) whose halo is ZERO: the kernel reads y on the tile grown by the PSF radius.
 * `x_a`, `x_b` are padded working volumes (allocation start, zero halos, written only inside the
 * logical window). x_a holds the initial estimate unless `init_from_y` (x0 = y, read in place).
 * Iteration i reads (i even ? x_a : x_b) and writes the other; the last one writes the dense
 * (Z, Y, X) `x_out` instead when that is not NULL. With x_out == NULL the result is in x_b for odd
 * `iters`, in x_a for even.
 */
int lsr_rl_sep_fused_supported(int pz, int py, int px);
/* `taps`: a DEVICE array of lsr_rl_sep_fused_taps_count() floats; lsr_rl_sep_fused_prepare_taps
 * fills the HOST image from the three PSF factors (host arrays of pz / py / px taps), the caller
 * uploads it once per PSF. The kernel reads it through the scalar cache. */
int lsr_rl_sep_fused_taps_count(void);
int lsr_rl_sep_fused_prepare_taps(const float* kz_host, int pz, const float* ky_host, int py,
                                  const float* kx_host, int px, float* taps_host);
int lsr_rl_sep_fused_f32(const float* y, int64_t y_pitch, int64_t y_plane, int init_from_y,
                         float* x_a, float* x_b, float* x_out, int64_t Z, int64_t Y, int64_t X,
                         const float* taps, int pz, int py, int px, const float* nz,
                         const float* ny, const float* nx, int iters, float eps,
                         lsr_stream_t stream);

/*
 * The same one-launch iteration for PSFs that separate along y only, psf[a][b][c] = ky[b] * kzx[a][c]
 * (rl_fused_ysep.hip) -- the PSF of an oblique light sheet, tilted in (z, x) and Gaussian along y; SURVEY.md
 * section 8(d)'s secondary PSF.  Volumes as lsr_rl_sep_fused_f32 (y padded with a zero halo, x_a / x_b padded
 * working volumes, x_out dense or NULL).  `taps`: a DEVICE array of lsr_rl_ysep_fused_taps_count() floats whose
 * HOST image lsr_rl_ysep_fused_prepare_taps fills from ky (py taps) and kzx (pz x px, C order);
 * norm_table / norm_full: the full PSF's, as for lsr_correlate_dense_f32.  Compiled for pz <= 11 and
 * py, px <= 9 (lsr_rl_ysep_fused_supported; otherwise LSR_E_UNSUPPORTED: use lsr_correlate_zxy_padded_f32 twice
 * per iteration).  Results are bit-identical to that two-launch form.
 */
int lsr_rl_ysep_fused_supported(int pz, int py, int px);
int lsr_rl_ysep_fused_taps_count(void);
int lsr_rl_ysep_fused_prepare_taps(const float* ky_host, int py, const float* kzx_host, int pz, int px,
                                   float* taps_host);
int lsr_rl_ysep_fused_f32(const float* y, int64_t y_pitch, int64_t y_plane, int init_from_y, float* x_a,
                          float* x_b, float* x_out, int64_t Z, int64_t Y, int64_t X, const float* taps,
                          int pz, int py, int px, const double* norm_table, float norm_full, int iters,
                          float eps, lsr_stream_t stream);

int lsr_rl_dense_f32(const float* y, float* x, float* ratio, int64_t Z, int64_t Y, int64_t X,
                     const float* psf, const float* psf_flipped, int pz, int py, int px,
                     const double* norm_table, int iters, float eps, lsr_stream_t stream);

/*
 * Tuned dense (non-separable) path: fp32-VALU-bound stencil on padded volumes, for PSFs with
 * pz <= 11 and py, px <= 9 (anything else: LSR_E_UNSUPPORTED, use the generic dense entries).
 *
 * The taps are prepared ONCE on the host and uploaded by the caller:
 *   n = lsr_dense_taps_count(pz, py, px)                    number of floats (or < 0)
 *   lsr_dense_prepare_taps(psf_host, pz, py, px, flip, taps_host)   HOST pointers, no GPU work;
 *       flip = 0: correlation with the PSF (H^T), flip = 1: with the reversed PSF (H = convolution)
 * and `taps` below is that array in DEVICE memory. `in` lives in a padded volume
 * (lsr_sep_padded_shape); `norm_table` as for lsr_correlate_dense_f32, `norm_full` = sum of taps.
 */
int lsr_dense_taps_count(int pz, int py, int px);
int lsr_dense_prepare_taps(const float* psf_host, int pz, int py, int px, int flip,
                           float* taps_host);

int lsr_correlate_dense_padded_f32(const float* in, int64_t in_pitch, int64_t in_plane,
                                   const float* aux, int64_t aux_pitch, int64_t aux_plane,
                                   float* out, int64_t out_pitch, int64_t out_plane, int64_t Z,
                                   int64_t Y, int64_t X, const float* taps, int pz, int py, int px,
                                   int epilogue, float eps, const double* norm_table,
                                   float norm_full, lsr_stream_t stream);

/*
 * One launch for a PSF that separates along y, psf[z][y][x] = ky[y] * kzx[z][x] (the tilted
 * light-sheet PSF): out = epilogue(correlate(in, psf), aux) with LSR_EPI_RATIO or LSR_EPI_UPDATE.
 * `taps_zx` = lsr_dense_prepare_taps of the pz x py x px array that holds kzx in its centre y row and
 * zeros elsewhere (flip = 1 for H x); `ky` = the py y taps on the device (reversed for H x);
 * norm_table / norm_full = those of the full PSF.  pz*px + py FMAs per voxel instead of pz*py*px.
 * Volumes and strides as lsr_correlate_dense_padded_f32.
 */
int lsr_correlate_zxy_padded_f32(const float* in, int64_t in_pitch, int64_t in_plane,
                                 const float* aux, int64_t aux_pitch, int64_t aux_plane, float* out,
                                 int64_t out_pitch, int64_t out_plane, int64_t Z, int64_t Y, int64_t X,
                                 const float* taps_zx, const float* ky, int pz, int py, int px,
                                 int epilogue, float eps, const double* norm_table, float norm_full,
                                 lsr_stream_t stream);
int lsr_rl_dense_padded_f32(const float* y, int64_t y_pitch, int64_t y_plane, int init_from_y,
                            float* x_pad, float* ratio_pad, float* x_out, int64_t Z, int64_t Y, int64_t X,
                            const float* taps, const float* taps_flipped, int pz, int py, int px,
                            const double* norm_table, float norm_full, int iters, float eps,
                            lsr_stream_t stream);

/*
 * Axial PSFs beyond 15 taps: 1-D correlation along z with up to lsr_correlate_z_max_taps() (31) odd taps and the RL
 * epilogues (csrc/correlate_z.hip) -- the z half of a separable PSF whose in-plane factors run through
 * lsr_correlate_sep_strided_f32 with ONE z tap: H x = Cz(Cyx(x)), two launches per correlation, four per iteration.
 * Any strides (floats; rows may be padded), no halo needed, `in` != `out`; `out` may alias `aux`.  UPDATE divides by
 * nz[z] * ny[y] * nx[x] and adds the launch's RL scalars to stats[0..2] when that is not NULL (see below).
 */
int lsr_correlate_z_max_taps(void);
int lsr_correlate_z_f32(const float* in, int64_t in_pitch, int64_t in_plane, const float* aux, int64_t aux_pitch,
                        int64_t aux_plane, float* out, int64_t out_pitch, int64_t out_plane, int64_t Z, int64_t Y,
                        int64_t X, const float* wz, int pz, int epilogue, float eps, const float* nz, const float* ny,
                        const float* nx, double* stats, lsr_stream_t stream);

/*
 * Richardson-Lucy reduction scalars (the north-star's "wavefront reductions for the ratio / normalisation").
 * Every entry that finishes an RL iteration has a `_stats` form with one more argument, `double* stats` (DEVICE memory;
 * host memory for the *_cpu twins); NULL = the plain entry, the same kernels, no cost.  Per iteration i three sums over
 * the voxels of the volume, formed in the epilogue that writes x_{i+1} (no extra pass, no extra HBM byte):
 *
 *   stats[3 i + 0]  flux    sum x_i * H^T(ratio_i)  ==  sum x_{i+1} * H^T 1   -- what the multiplicative update
 *                           conserves: equals sum_v y_v (H x_i)_v / ((H x_i)_v + eps), i.e. sum y as eps -> 0
 *   stats[3 i + 1]  change  sum |x_{i+1} - x_i|                                -- the update norm
 *   stats[3 i + 2]  total   sum x_{i+1}
 *
 * change / total is the relative change a caller can stop on (the Python host: richardson_lucy(..., tol=)).  Per thread
 * the sums run in f32 over the voxels it owns, then wavefront reduction (DPP) -> LDS -> one f64 atomic add per workgroup
 * and sum; the only run-to-run freedom is the order of those f64 adds (~1e-16 relative).
 *   lsr_rl_*_stats_f32:        stats = 3 * iters doubles, zeroed by the entry on `stream` before the first launch;
 *   lsr_correlate_*_stats_f32: stats = 3 doubles the launch ADDS to (LSR_EPI_UPDATE only; ignored for other epilogues):
 *                              the caller zeroes them -- these are the per-launch building blocks of the loops above.
 * Otherwise arguments, limits and results are exactly those of the entry without `_stats`.
 */
int lsr_rl_sep_fused_stats_f32(const float* y, int64_t y_pitch, int64_t y_plane, int init_from_y,
                               float* x_a, float* x_b, float* x_out, int64_t Z, int64_t Y, int64_t X,
                               const float* taps, int pz, int py, int px, const float* nz,
                               const float* ny, const float* nx, int iters, float eps, double* stats,
                               lsr_stream_t stream);
int lsr_rl_sep_stats_f32(const float* y, int64_t y_pitch, int64_t y_plane, int init_from_y,
                         float* x_pad, float* ratio_pad, float* x_out, int64_t Z, int64_t Y, int64_t X,
                         const float* kz, const float* kz_flipped, int pz,
                         const float* ky, const float* ky_flipped, int py, const float* kx,
                         const float* kx_flipped, int px, const float* nz, const float* ny,
                         const float* nx, int iters, float eps, double* stats, lsr_stream_t stream);
int lsr_rl_ysep_fused_stats_f32(const float* y, int64_t y_pitch, int64_t y_plane, int init_from_y, float* x_a,
                                float* x_b, float* x_out, int64_t Z, int64_t Y, int64_t X, const float* taps,
                                int pz, int py, int px, const double* norm_table, float norm_full, int iters,
                                float eps, double* stats, lsr_stream_t stream);
int lsr_rl_dense_padded_stats_f32(const float* y, int64_t y_pitch, int64_t y_plane, int init_from_y,
                                  float* x_pad, float* ratio_pad, float* x_out, int64_t Z, int64_t Y, int64_t X,
                                  const float* taps, const float* taps_flipped, int pz, int py, int px,
                                  const double* norm_table, float norm_full, int iters, float eps, double* stats,
                                  lsr_stream_t stream);
int lsr_rl_dense_stats_f32(const float* y, float* x, float* ratio, int64_t Z, int64_t Y, int64_t X,
                           const float* psf, const float* psf_flipped, int pz, int py, int px,
                           const double* norm_table, int iters, float eps, double* stats, lsr_stream_t stream);
int lsr_correlate_sep_stats_f32(const float* in, float* out, const float* aux, int64_t Z, int64_t Y,
                                int64_t X, const float* wz, int pz, const float* wy, int py,
                                const float* wx, int px, int epilogue, float eps, const float* nz,
                                const float* ny, const float* nx, double* stats, lsr_stream_t stream);
int lsr_correlate_dense_stats_f32(const float* in, float* out, const float* aux, int64_t Z, int64_t Y,
                                  int64_t X, const float* w, int pz, int py, int px, int epilogue,
                                  float eps, const double* norm_table, double* stats, lsr_stream_t stream);
int lsr_correlate_sep_strided_stats_f32(const float* in, int64_t in_pitch, int64_t in_plane,
                                        const float* aux, int64_t aux_pitch, int64_t aux_plane,
                                        float* out, int64_t out_pitch, int64_t out_plane, int64_t Z,
                                        int64_t Y, int64_t X, const float* wz, int pz, const float* wy,
                                        int py, const float* wx, int px, int epilogue, float eps,
                                        const float* nz, const float* ny, const float* nx, double* stats,
                                        lsr_stream_t stream);
int lsr_correlate_dense_padded_stats_f32(const float* in, int64_t in_pitch, int64_t in_plane,
                                         const float* aux, int64_t aux_pitch, int64_t aux_plane,
                                         float* out, int64_t out_pitch, int64_t out_plane, int64_t Z,
                                         int64_t Y, int64_t X, const float* taps, int pz, int py, int px,
                                         int epilogue, float eps, const double* norm_table,
                                         float norm_full, double* stats, lsr_stream_t stream);
int lsr_correlate_zxy_padded_stats_f32(const float* in, int64_t in_pitch, int64_t in_plane,
                                       const float* aux, int64_t aux_pitch, int64_t aux_plane, float* out,
                                       int64_t out_pitch, int64_t out_plane, int64_t Z, int64_t Y, int64_t X,
                                       const float* taps_zx, const float* ky, int pz, int py, int px,
                                       int epilogue, float eps, const double* norm_table, float norm_full,
                                       double* stats, lsr_stream_t stream);

/*
 * Reductions and filters behind the DynaTrack shift estimators that run on the deskewed volume
 * (shrimpy/dynatrack/tracking.py). `scratch` = lsr_reduce_scratch_bytes() bytes of
 * device memory; outputs are device arrays; sums are fp64, reduced in a fixed order.
 *
 *   lsr_minmax_f32            out2 = {min, max} of n floats                         (:533-535, :583)
 *   lsr_histogram_f32         torch.histc(in, nbins, vmin, vmax) as uint32 counts   (:465, :586)
 *                             (n < 2^32 per call, LSR_E_UNSUPPORTED otherwise: add up pieces)
 *   lsr_weighted_centroid_f32 out4 = {sum w, sum w z, sum w y, sum w x}, w = max(v - background, 0)
 *                             (_intensity_center_of_mass, :596-649)
 *   lsr_mask_centroid_f32     the same with w = (v > threshold)  (_center_of_mass of a mask, :545-569)
 *   lsr_blur_reflect_f32      one axis (0 = z, 1 = y, 2 = x) of _gaussian_blur_3d (:386-422):
 *                             correlation with 2*radius+1 device taps, reflect borders
 *                             (index -k -> k; radius < axis length, radius <= 64); div != 0 maps
 *                             the input through (v - sub) / div first (the [0, 1] rescale, :533-535)
 */
int lsr_reduce_scratch_bytes(void);
int lsr_minmax_f32(const float* in, int64_t n, float* out2, void* scratch, lsr_stream_t stream);
int lsr_histogram_f32(const float* in, int64_t n, float vmin, float vmax, int nbins, unsigned* counts,
                      lsr_stream_t stream);
int lsr_weighted_centroid_f32(const float* in, int64_t Z, int64_t Y, int64_t X, float background,
                              double* out4, void* scratch, lsr_stream_t stream);
int lsr_mask_centroid_f32(const float* in, int64_t Z, int64_t Y, int64_t X, float threshold,
                          double* out4, void* scratch, lsr_stream_t stream);
int lsr_blur_reflect_f32(const float* in, float* out, int64_t Z, int64_t Y, int64_t X, int axis,
                         const float* taps, int radius, float sub, float div, lsr_stream_t stream);
/*
 * Element-wise steps of _phase_cross_corr (:266-378); the FFTs between them are library calls.
 *   lsr_match_shape_f32       _match_shape: per axis reflect-pad (left = d / 2) or centre-crop to the FFT shape
 *   lsr_cross_power_c64       a <- a * conj(b) over n complex64 values (interleaved re, im)
 *   lsr_peak_abs_shifted_f32  argmax(fftshift(|in|)) as a flat index into the SHIFTED (Z, Y, X) array,
 *                             first maximum in that order (torch.argmax), without writing either
 */
int lsr_match_shape_f32(const float* in, int64_t Zi, int64_t Yi, int64_t Xi, float* out, int64_t Zo,
                        int64_t Yo, int64_t Xo, lsr_stream_t stream);
int lsr_cross_power_c64(float* a, const float* b, int64_t n, lsr_stream_t stream);
/* dst[a][c][b] = src[a][b][c] for complex64 (8-byte) elements, out of place: the layout change between
 * the per-axis transforms of the cross-correlation's 3-D FFT (A = 1: a plain 2-D transpose). */
int lsr_transpose_last2_c64(const float* src, float* dst, int64_t A, int64_t B, int64_t C, lsr_stream_t stream);
/*
 * The z leg of the cross-correlation in one pass: g <- N * IFFT_z( f1 * conj( FFT_z(g) ) ) along the first axis of
 * g ([N][XC][Y] complex64: the moving volume's spectrum after its x and y transforms), f1 = the reference's fully
 * transformed spectrum in [XC][Y][N] (z contiguous), twiddles[k] = exp(-2 pi i k / N) (N complex64, device).  Replaces
 * transpose + forward z transform + cross power + inverse z transform + transpose (csrc/zcorr.hip).  N: 5-smooth, 2..256
 * (lsr_cross_correlate_z_supported; LSR_E_UNSUPPORTED otherwise -- callers keep the five-pass route).
 */
int lsr_cross_correlate_z_supported(int64_t n);
int lsr_cross_correlate_z_c64(const float* f1, float* g, const float* twiddles, int64_t N, int64_t Y, int64_t XC,
                              lsr_stream_t stream);
/*
 * The x leg of the same cross-correlation, each way in one kernel (csrc/rfft_rows.hip).  Row lengths X that are
 * multiples of 4 with X / 2 5-smooth and <= 2048 (lsr_rfft_rows_supported); tw_half[k] = exp(-2 pi i k / (X/2)),
 * k < X / 4, and tw_x[k] = exp(-2 pi i k / X), k <= X / 2 (complex64, device).
 * lsr_rfft_rows_t_c64: spec[z][k][y] = sum_n v(z, y, n) exp(-2 pi i k n / X), k <= X / 2, where v is `in`
 *   ((Zi, Yi, Xi) float32) reflect-padded / centre-cropped to the grid (Z, Y, X) the way _match_shape does
 *   (tracking.py:266-306: pad left = d // 2, crop start = d // 2) -- the padded volume is never written.
 *   Replaces _match_shape + the real-to-complex transform along x + the transpose to y-contiguous.
 * lsr_irfft_rows_peak: the flat index of max |corr| in fftshift order (ties: the smallest), corr = the
 *   unnormalised complex-to-real inverse along x of spec ([Z][X/2+1][Y]); corr is never written.  `scratch`:
 *   lsr_rfft_rows_scratch_bytes(Z, Y) bytes.  Replaces the transpose back, irfft along x, fftshift(abs()), argmax.
 */
int lsr_rfft_rows_supported(int64_t n);
int64_t lsr_rfft_rows_scratch_bytes(int64_t Z, int64_t Y);
int lsr_rfft_rows_t_c64(const float* in, int64_t Zi, int64_t Yi, int64_t Xi, float* spec, int64_t Z, int64_t Y,
                        int64_t X, const float* tw_half, const float* tw_x, lsr_stream_t stream);
int lsr_irfft_rows_peak(const float* spec, int64_t Z, int64_t Y, int64_t X, const float* tw_half, const float* tw_x,
                        long long* out_index, void* scratch, lsr_stream_t stream);
/*
 * Richardson-Lucy in the Fourier domain (shrimpy_amd/deconvolve_fft.py): PSFs beyond the stencil kernels' extents --
 * a measured bead PSF is a dense 15 x 18 x 18 ... 30 x 36 x 18 voxel patch (/root/reference/scripts/measure_psf.py:187-190)
 * -- cost the same as any other there.  No reference code for RL itself (/root/reference/docs/data_structure.md:58-62);
 * the arithmetic is that of the stencil path: x <- x * H^T( y / (H x + eps) ) / H^T 1, zero-padded
 * borders, with H x and H^T r evaluated as products of spectra on a grid (Z, Y, X) >= volume + PSF radius per axis.
 * lsr_rfft_rows_zero_t_c64: as lsr_rfft_rows_t_c64 with the source ((Zi, Yi, Xi) <= grid) at the grid's origin and
 *   zeros behind it -- a linear convolution, not a circular one; tiles of padding rows (y >= Yi) in the source's planes
 *   store zeros without a transform, planes z >= Zi are NOT written (see z_valid below).
 * lsr_spectrum_multiply_z_c64: g <- N * IFFT_z( f1 * FFT_z(g) ) (conj_f1 = 0: H x) or N * IFFT_z( conj(f1) * FFT_z(g) )
 *   (conj_f1 = 1: H^T r); layouts and lengths as lsr_cross_correlate_z_c64, f1 = the PSF's spectrum (its centre tap at
 *   the grid's origin, wrapped).  z_valid: planes [z_valid, N) of g are the zero padding behind the volume -- taken as
 *   zeros and never read, so the forward x and y legs need not produce them; z_keep: only planes [0, z_keep) of the
 *   result are stored -- the inverse y and x legs read no others (N, N = everything).
 * lsr_irfft_rows_rl_f32: the inverse x leg with the iteration's epilogue.  v = scale * (complex-to-real inverse of
 *   spec along x) on the grid's first (Zo, Yo, Xo) points, scale = 1 / (Z Y X);
 *     LSR_EPI_RATIO:  out = aux / (max(v, 0) + eps)    aux = y
 *     LSR_EPI_UPDATE: out = aux * v / H^T 1            aux = x; H^T 1 = norm_full inside, from norm_table ((pz+1)(py+1)
 *                     (px+1) prefix sums of the PSF, float64, device) within a PSF radius of the border; stats = 3 doubles
 *                     the launch ADDS the iteration's flux / change / total to (as lsr_correlate_*_stats_f32) or NULL.
 *   out may be aux (in place).  The convolved volume is never written.
 * lsr_rl_rows_chain_f32: the same, CHAINED into the forward x leg of the iteration's next convolution: the epilogue's
 *   output row is zero-padded, transformed and stored back over its tile of `spec`, which afterwards holds what
 *   lsr_rfft_rows_zero_t_c64(out) would have produced (tiles of pure padding: zeros).  LSR_EPI_RATIO: `out` may be NULL --
 *   the ratio then never exists in memory; LSR_EPI_UPDATE stores x_new once and nobody reads it back for the transform.
 */
int lsr_rfft_rows_zero_t_c64(const float* in, int64_t Zi, int64_t Yi, int64_t Xi, float* spec, int64_t Z, int64_t Y,
                             int64_t X, const float* tw_half, const float* tw_x, lsr_stream_t stream);
int lsr_spectrum_multiply_z_c64(const float* f1, float* g, const float* twiddles, int64_t N, int64_t Y, int64_t XC,
                                int conj_f1, int64_t z_valid, int64_t z_keep, lsr_stream_t stream);
int lsr_irfft_rows_rl_f32(const float* spec, int64_t Z, int64_t Y, int64_t X, const float* tw_half, const float* tw_x,
                          int epilogue, const float* aux, float* out, int64_t Zo, int64_t Yo, int64_t Xo, float scale,
                          float eps, int pz, int py, int px, const double* norm_table, float norm_full, double* stats,
                          lsr_stream_t stream);
int lsr_rl_rows_chain_f32(float* spec, int64_t Z, int64_t Y, int64_t X, const float* tw_half, const float* tw_x,
                          int epilogue, const float* aux, float* out, int64_t Zo, int64_t Yo, int64_t Xo, float scale,
                          float eps, int pz, int py, int px, const double* norm_table, float norm_full, double* stats,
                          lsr_stream_t stream);
/* b <- a * conj(b): the same product written over the second operand, so that `a` (the spectrum of
 * a reference volume that is compared against many timepoints) can be kept. */
int lsr_cross_power_into_c64(const float* a, float* b, int64_t n, lsr_stream_t stream);
int lsr_peak_abs_shifted_f32(const float* in, int64_t Z, int64_t Y, int64_t X, long long* out_index,
                             void* scratch, lsr_stream_t stream);

/*
 * Host twins (csrc/host_twins.hip): the same signatures with HOST pointers, the same argument checks and the
 * same arithmetic in the same order, so the results equal the device entry points' bit for bit.  They serve
 * the boxes where the reference itself resolves to the CPU (shrimpy/preprocessing.py:78-82 -- its CI has no
 * GPU, shrimpy/tests/conftest.py:11-17) and BASELINE config 1; they are product code and never touch oracle/.
 * `stream` is ignored.  Work is split over at most lsr_set_host_threads(n) plain threads (default 1, no OpenMP).
 * lsr_affine_f32_cpu takes LSR_MODE_CONSTANT / LSR_MODE_GRID_CONSTANT only (its arithmetic is always fp64).
 */
int lsr_set_host_threads(int n);
int lsr_get_host_threads(void);
int lsr_deskew_f32_cpu(const float* in, int64_t Z, int64_t Y, int64_t X, float* out, int64_t Zo,
                       int64_t Yo, int64_t Xo, int64_t out_pitch, int64_t out_plane, int64_t Zd,
                       const double M[12], int avg_n, lsr_stream_t stream);
int lsr_deskew_u16_cpu(const uint16_t* in, int64_t Z, int64_t Y, int64_t X, float* out, int64_t Zo,
                       int64_t Yo, int64_t Xo, int64_t out_pitch, int64_t out_plane, int64_t Zd,
                       const double M[12], int avg_n, lsr_stream_t stream);
int lsr_affine_f32_cpu(const float* in, int64_t Zi, int64_t Yi, int64_t Xi, float* out, int64_t Zo,
                       int64_t Yo, int64_t Xo, const double M[12], float cval, int mode,
                       lsr_stream_t stream);
int lsr_average_slices_f32_cpu(const float* in, int64_t Zd, int64_t Y, int64_t X, float* out,
                               int64_t Zo, int avg_n, lsr_stream_t stream);
int lsr_correlate_sep_f32_cpu(const float* in, float* out, const float* aux, int64_t Z, int64_t Y,
                              int64_t X, const float* wz, int pz, const float* wy, int py,
                              const float* wx, int px, int epilogue, float eps, const float* nz,
                              const float* ny, const float* nx, lsr_stream_t stream);
int lsr_correlate_dense_f32_cpu(const float* in, float* out, const float* aux, int64_t Z, int64_t Y,
                                int64_t X, const float* w, int pz, int py, int px, int epilogue,
                                float eps, const double* norm_table, lsr_stream_t stream);
int lsr_rl_dense_f32_cpu(const float* y, float* x, float* ratio, int64_t Z, int64_t Y, int64_t X,
                         const float* psf, const float* psf_flipped, int pz, int py, int px,
                         const double* norm_table, int iters, float eps, lsr_stream_t stream);
/* ... with the reduction scalars (see lsr_rl_*_stats_f32; `stats` is HOST memory, sums in f64 per row range) */
int lsr_correlate_sep_stats_f32_cpu(const float* in, float* out, const float* aux, int64_t Z, int64_t Y,
                                    int64_t X, const float* wz, int pz, const float* wy, int py,
                                    const float* wx, int px, int epilogue, float eps, const float* nz,
                                    const float* ny, const float* nx, double* stats, lsr_stream_t stream);
int lsr_correlate_dense_stats_f32_cpu(const float* in, float* out, const float* aux, int64_t Z, int64_t Y,
                                      int64_t X, const float* w, int pz, int py, int px, int epilogue,
                                      float eps, const double* norm_table, double* stats, lsr_stream_t stream);
int lsr_rl_dense_stats_f32_cpu(const float* y, float* x, float* ratio, int64_t Z, int64_t Y, int64_t X,
                               const float* psf, const float* psf_flipped, int pz, int py, int px,
                               const double* norm_table, int iters, float eps, double* stats,
                               lsr_stream_t stream);
int lsr_flatfield_pattern_f32_cpu(const float* in, int64_t Z, int64_t Y, int64_t X, float* pattern,
                                  float* mean_out, void* scratch /* unused */, lsr_stream_t stream);
int lsr_flatfield_pattern_u16_cpu(const uint16_t* in, int64_t Z, int64_t Y, int64_t X, float* pattern,
                                  float* mean_out, void* scratch /* unused */, lsr_stream_t stream);
int lsr_flatfield_apply_f32_cpu(const float* in, const float* pattern, const float* mean_dev,
                                float* out, int64_t Z, int64_t Y, int64_t X, lsr_stream_t stream);
int lsr_flatfield_apply_u16_cpu(const uint16_t* in, const float* pattern, const float* mean_dev,
                                float* out, int64_t Z, int64_t Y, int64_t X, lsr_stream_t stream);
/*
 * ... and of the DynaTrack estimators (csrc/estimators_host.hip), for a tracker that runs where the reference's own
 * does without a GPU (shrimpy/dynatrack/tracking.py:1054).  Identical values for min / max, the histogram, the shape
 * map, the cross power, the peak and the blur (the kernels' FMA chain); the centroid sums are fp64 like the kernels'
 * but added in row order (equal to the last bits of a double).  `scratch` is unused.  The FFTs between the
 * cross-correlation's steps stay library calls (torch.fft on the CPU).
 */
int lsr_minmax_f32_cpu(const float* in, int64_t n, float* out2, void* scratch, lsr_stream_t stream);
int lsr_histogram_f32_cpu(const float* in, int64_t n, float vmin, float vmax, int nbins, unsigned* counts,
                          lsr_stream_t stream);
int lsr_weighted_centroid_f32_cpu(const float* in, int64_t Z, int64_t Y, int64_t X, float background,
                                  double* out4, void* scratch, lsr_stream_t stream);
int lsr_mask_centroid_f32_cpu(const float* in, int64_t Z, int64_t Y, int64_t X, float threshold,
                              double* out4, void* scratch, lsr_stream_t stream);
int lsr_blur_reflect_f32_cpu(const float* in, float* out, int64_t Z, int64_t Y, int64_t X, int axis,
                             const float* taps, int radius, float sub, float div, lsr_stream_t stream);
int lsr_match_shape_f32_cpu(const float* in, int64_t Zi, int64_t Yi, int64_t Xi, float* out, int64_t Zo,
                            int64_t Yo, int64_t Xo, lsr_stream_t stream);
int lsr_cross_power_c64_cpu(float* a, const float* b, int64_t n, lsr_stream_t stream);
int lsr_cross_power_into_c64_cpu(const float* a, float* b, int64_t n, lsr_stream_t stream);
int lsr_peak_abs_shifted_f32_cpu(const float* in, int64_t Z, int64_t Y, int64_t X, long long* out_index,
                                 void* scratch, lsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LSRECON_H */
