// spmm_kernels.h — internal interface between the C ABI (capi.cpp) and the HIP
// kernel translation units. Not installed; the public surface is include/gespmm.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace gespmm {

constexpr int kThreads = 256;  // workgroup = 4 wavefronts of 64 lanes
constexpr int kWaves = 4;
constexpr int kTile = 64;      // CSR entries staged in LDS per wavefront refill
constexpr int kSlabRowsPerGroup = 8; // slab-blocked path: rows a lane group walks per launch
constexpr int kMaxRowsPerWave = 32;  // streaming kernel: rows owned by one wavefront

constexpr int kReduceSum = 0;
constexpr int kReduceMax = 1;

constexpr int kFlagNoXcdRemap = 0x1;  // == GESPMM_FLAG_NO_XCD_REMAP
constexpr int kFlagNtStore = 0x2;     // == GESPMM_FLAG_NT_STORE
constexpr int kFlagForceIdx64 = 0x4;  // == GESPMM_FLAG_FORCE_IDX64
constexpr int kFlagShallowUnroll = 0x10; // == GESPMM_FLAG_SHALLOW_UNROLL
constexpr int kFlagBatchStream = 0x20;   // == GESPMM_FLAG_BATCH_STREAM (force the batch-stream kernel)
constexpr int kFlagStrictOrder = 0x100;  // == GESPMM_FLAG_STRICT_ORDER (never split long rows)
constexpr int kFlagSplitLongRows = 0x200; // == GESPMM_FLAG_SPLIT_LONG_ROWS (always run the long-row pass)
constexpr int kLongRowThreshold = 2048;  // entries; lower bound of the long-row threshold (32 x mean degree)
constexpr int64_t kMaxGridBlocks = (1ll << 24) - 1;  // 2^32 threads per launch / 256 threads per workgroup
constexpr int kLongRowChunk = 2048;      // entries of a long row one workgroup sums per partial row
constexpr int64_t kLongRowMinNnz = 1 << 23;  // auto: only matrices this large get the long-row pass
constexpr int kFlagSlabBlocked = 0x400;   // == GESPMM_FLAG_SLAB_BLOCKED (force the cache-blocked path)
constexpr int kFlagNoSlabBlocked = 0x800; // == GESPMM_FLAG_NO_SLAB_BLOCKED
constexpr int kFlagReuseSplit = 0x2000;   // == GESPMM_FLAG_REUSE_SPLIT
constexpr int kFlagAllowReassoc = 0x1000; // == GESPMM_FLAG_ALLOW_REASSOCIATION
constexpr int kFlagSc1Store = 0x8000;     // C stores with system scope (written through, not kept in the XCD's L2)
constexpr int kFlagSegStream = 0x80;     // == GESPMM_FLAG_SEG_STREAM (force the segmented-stream kernel)

struct SpmmArgs {
    const int32_t* rowptr;
    const int32_t* colind;
    const float* val;  // nullptr: A == 1 on its pattern
    const float* B;
    float* C;
    int32_t M;
    int32_t N;
    int32_t nblk;   // row blocks (filled in by the launcher)
    int32_t ntile;  // column tiles (filled in by the launcher)
    int32_t flags;
    int32_t rpw;    // streaming kernel: rows per wavefront (filled in by the launcher)
    const int32_t* row_begin;  // slab-blocked path: per-row CSR range of the current slab
    const int32_t* row_end;
    int32_t accumulate;        // slab-blocked path: 0 = first slab (C = ...), 1 = C += ...
    int32_t long_row;  // > 0: rows with more entries are skipped by the main kernel (long-row kernel does them)
    // long-row pass workspace (device): the main kernel appends the rows it skips to these lists
    int32_t* lr_hdr;     // {nchunks, nrows, -, -}
    int32_t* lr_rows;    // int4 per long row: {row, first chunk slot, #chunks, -}
    int32_t* lr_chunks;  // int2 per chunk: {row, chunk index}
    int32_t lr_chunk, lr_max_rows, lr_max_chunks;
    float empty;    // max reducer: value of rows without non-zeros / initial accumulator
    // plan mode (plan.cpp): rowptr/colind/val are the plan's row-permuted copy of the matrix, wavefront w works on
    // task w = int4 {first permuted row, #rows (<= kMaxRowsPerWave), CSR begin, CSR end}, and row i of the
    // permuted matrix is written to C row perm[i]
    const int32_t* tasks;
    const int32_t* perm;
    int32_t ntasks;
    // the same for the segmented-stream kernel, whose unit of work is a lane GROUP: group q works on gtasks[q]
    const int32_t* gtasks;
    int32_t ngtasks;
    // launch guard (auto_plan.cpp, asynchronous mode): when set, every workgroup first reads *guard and leaves unless it equals guard_want
    // — a fingerprint kernel earlier on the stream decided whether the cached plan still describes the caller's arrays. NULL: no check.
    const int32_t* guard;
    int32_t guard_want;
};

// A device word + the value a guarded launch runs for (run_spmm / the plan's launch: only paths that are ONE kernel take a guard).
struct LaunchGuard {
    const int32_t* word;
    int32_t want;
};
constexpr int kNotGuardable = -100;  // internal: the launch would be several kernels (long-row pass, cache blocking, ...): no guard

// Launch geometry resolved by the host-side selector (select.cpp).
struct Geometry {
    int vec;      // V: floats per lane per strip (1, 2, 4)
    int strips;   // S: strips per lane (1, 2)
    int group;    // W: lanes per row (4..64)
    int rows_per_wave;  // batch-stream kernel: consecutive rows owned by one wavefront
    int rows_per_group; // segmented-stream kernel: consecutive rows owned by one lane group
    bool idx64;   // 64-bit byte offsets into B
    bool segmented;  // segmented-stream kernel (else batch-stream)
    bool slab_blocked;     // dense graph: one launch per column slab (cache blocking)
    int slab_rows;         // B rows per slab
    int64_t K;             // columns of A (for the slab count)
    bool split_long_rows;  // run the long-row pass
    int long_row_threshold;  // rows with more entries than this go to the long-row pass
    int reduce;   // kReduceSum / kReduceMax
    bool sc1_store;  // C stored with system scope (B is L2-resident and C is not: keep C lines out of the L2)
};

hipError_t launch_spmm_naive(const SpmmArgs& a, const Geometry& geo, hipStream_t st);
hipError_t launch_spmm_stream(const SpmmArgs& a, const Geometry& geo, hipStream_t st);
hipError_t launch_spmm_segstream(const SpmmArgs& a, const Geometry& geo, hipStream_t st);
// spmm_stream_plan.hip: the same two kernels on a plan's task table (a.tasks / a.gtasks + a.perm)
hipError_t launch_spmm_stream_planned(const SpmmArgs& a, const Geometry& geo, hipStream_t st);
hipError_t launch_spmm_segstream_planned(const SpmmArgs& a, const Geometry& geo, hipStream_t st);
// The two paths below need a temporary: the caller's (ext_ws, 16-byte aligned, >= *_workspace_bytes) or,
// when that is absent or too small, a stream-ordered block from the library's pool (workspace.h).
size_t longrows_workspace_bytes(int64_t nnz, int64_t N, int long_row);
hipError_t launch_spmm_stream_with_longrows(SpmmArgs a, const Geometry& geo, int64_t nnz, void* ext_ws, size_t ext_bytes,
                                            hipStream_t st);
size_t slabblocked_workspace_bytes(int64_t M, const Geometry& geo);
hipError_t launch_spmm_slabblocked(const SpmmArgs& a, const Geometry& geo, void* ext_ws, size_t ext_bytes,
                                   hipStream_t st);
hipError_t launch_spmm_parreduce(const SpmmArgs& a, const Geometry& geo, hipStream_t st);

// spmm_staged.hip — scalar-stream kernel with a block's most used B rows staged in LDS (clustered plans, N = 128 / 256 and, as
// 256-column tiles bound to XCDs, 512 / 1024; sum).
// plan_device.hip (device_build_staging) writes the tables: blocks of staged_block_rows(N) consecutive rows of the clustered matrix,
// `waves` tasks per block (int4 {first row, #rows, stream begin, stream end}), per block the staged columns, and the record stream
// `ev`: per row its entries {code, value bits} (bit 31 of the code clear: column; set: staged slot in its low bits) followed by ONE
// row-end record {kStagedRowEnd, C row of that row} — also for rows without entries — so a wavefront needs nothing but its stream
// range: no row pointers, no row ids, no position compares on the walk (round 5). Record position of entry p of row r: p + r; the
// stream is padded by kStagedPad records.
constexpr int kStagedMaxWaves = 16;      // wavefronts per block: 4, 8 or 16 (StagedShape)
constexpr int kStagedLdsPerWave = 4096;  // bytes of staged B rows per wavefront of the block (16 wavefronts: 64 KB); GESPMM_STAGED_LDS_KB=8: 8192
constexpr int kStagedPad = 64;
constexpr int kStagedRowEnd = 0x40000000;  // code of a row-end record (both address shifts of the kernel drop the bit: slot 0 / column 0)
constexpr int kStagedMaxRow = 2048;  // longer rows are walked by the streaming kernel's long-row pass, not by one wavefront
inline bool staged_stream_fits(int64_t M, int64_t nnz) { return nnz + M + kStagedPad < (1ll << 31) - 64; }  // 32-bit stream positions
struct StagedArgs {
    const int32_t* rowptr;    // clustered matrix (not read by the kernel since round 5: the stream carries the row ends)
    const int32_t* ev;        // 2 * (nnz + M + kStagedPad) words
    const int32_t* perm;      // C row of clustered row i (not read by the kernel since round 5: the row-end records carry it)
    const int32_t* tasks;     // nblocks * waves int4
    const int32_t* hot_cols;  // nblocks * H (H = staged_shape(N).slots)
    const int32_t* nhot;      // nblocks
    const float* B;
    float* C;
    int32_t nblocks;
    int32_t waves;            // wavefronts (= tasks) per block the tables were built for
    int32_t slots;            // staged rows per block the tables were built for (H: decides the LDS per wavefront, 4 or 8 KB)
    int32_t debug;            // experiments only (GESPMM_STAGED_DEBUG): 1 = skip the staging copy, 2 = every gather from LDS — WRONG results;
                              // 4 = phase clocks summed into dbg_clk
    unsigned long long* dbg_clk;
    // spmm_staged_gen.hip (filled in by its launcher): width, column tiles of 64 * VEC columns, value of rows without entries (max reducer)
    int32_t n;
    int32_t ntiles;
    float empty;
    const int32_t* guard;  // launch guard, as in SpmmArgs (NULL: no check)
    int32_t guard_want;
    // column-slab tables (plan.cpp: build_slab_tables; spmm_staged.hip only): the launch covers blocks blk0 .. blk0 + nblocks - 1 of the
    // tables; acc != 0: rows continue from the partial sums in C (task word 0 = C row of the task's first row, row-end codes carry the next)
    int32_t blk0;
    int32_t acc;
};
// Shape of a block at width N: `waves` wavefronts (0 = width not served) own `rows` consecutive rows of the clustered matrix and
// stage up to `slots` B rows (waves x 4 KB of LDS). GESPMM_STAGED_WAVES / GESPMM_STAGED_ROWS override it for experiments.
struct StagedShape {
    int waves, rows, slots;
};
StagedShape staged_shape(int64_t N);
inline int staged_rows_per_block_lds(int64_t N) { return staged_shape(N).slots; }  // H for this width; 0 = width not served
inline int staged_block_rows(int64_t N) { return staged_shape(N).rows; }
bool staged_serves(int64_t M, int64_t K, int64_t N);  // width served, B and C addressable (32-bit offsets; two 4 GB halves for the tiled widths)
hipError_t launch_spmm_staged(const StagedArgs& a, int64_t M, int64_t K, int64_t N, hipStream_t st);
// spmm_staged_narrow.hip — the same tables at N = 16 / 32 / 64: a wavefront is G = 64 / (N / 4) lane groups, each walking its own range
// of the record stream (`waves` x G tasks per block); staged rows from LDS and memory rows under complementary EXEC masks.
StagedShape staged_narrow_shape(int64_t N);  // waves = 0: width not served
int staged_narrow_groups(int64_t N);         // lane groups (tasks) per wavefront
bool staged_narrow_serves(int64_t M, int64_t K, int64_t N);
hipError_t launch_spmm_staged_narrow(const StagedArgs& a, int64_t M, int64_t K, int64_t N, hipStream_t st);
// spmm_staged_gen.hip — the same kernel for ANY width and for the max reducer (round 6): VEC = staged_gen_vec(N) floats per lane, column
// tiles of 64 * VEC columns, lanes past N masked at the row-end store, row strides by a scalar multiply. Tables: 16 tasks per block,
// 80 KB / (256 * VEC) slots (for N = 128 and 256 * 2^t the shapes of spmm_staged.hip: one set of tables serves its sum kernel and this
// file's max kernel).
int staged_gen_vec(int64_t N);  // 1, 2 or 4 (0: N < 1)
StagedShape staged_gen_shape(int64_t N);
bool staged_gen_serves(int64_t M, int64_t K, int64_t N);
hipError_t launch_spmm_staged_gen(const StagedArgs& a, int64_t M, int64_t K, int64_t N, int reduce, float empty, hipStream_t st);
// Which of the three kernels walks a plan's tables at width N: 1 = lane groups (N = 16 / 32 / 64, sum), 2 = spmm_staged.hip's tuned shapes
// (N = 128, 256 * 2^t, sum), 3 = the general kernel (every other width; the max reducer at every width but 16 / 32 / 64), 0 = none.
enum { kStagedNone = 0, kStagedNarrow = 1, kStagedTuned = 2, kStagedGeneral = 3 };
inline int staged_kernel_class(int64_t M, int64_t K, int64_t N, int reduce = kReduceSum) {
    if (staged_narrow_shape(N).waves) return (reduce == kReduceSum && staged_narrow_serves(M, K, N)) ? kStagedNarrow : kStagedNone;
    if (reduce == kReduceSum && staged_serves(M, K, N)) return kStagedTuned;
    return staged_gen_serves(M, K, N) ? kStagedGeneral : kStagedNone;
}
// Columns of one tile of the wide kernels at width N (128 / 256: the tile classes the policy's thresholds are measured for; 64: odd widths)
inline int staged_tile_class(int64_t N) { return 64 * staged_gen_vec(N); }
// any kernel: the block shape of width N (waves = 0: none serves it) and the tasks per block
inline StagedShape staged_shape_any(int64_t N) {
    const StagedShape w = staged_shape(N);
    if (w.waves) return w;
    const StagedShape nw = staged_narrow_shape(N);
    return nw.waves ? nw : staged_gen_shape(N);
}
inline int staged_tasks_per_block(int64_t N) {
    const StagedShape w = staged_shape(N);
    if (w.waves) return w.waves;
    const StagedShape nw = staged_narrow_shape(N);
    if (nw.waves) return nw.waves * staged_narrow_groups(N);
    return staged_gen_shape(N).waves;
}
inline bool staged_serves_any(int64_t M, int64_t K, int64_t N) { return staged_kernel_class(M, K, N) != kStagedNone; }

// spmm_records.hip — the padded-record kernel (round 6): narrow widths (4 <= N <= 64), short rows, sum reducer, plans only. A row
// is cut into pieces of kRecordPiece padded entry slots; a wavefront is 64 / W chains (W = records_group(N) lanes each) and consumes one
// BATCH = one piece per chain per step (headers + entries: one coalesced load per lane), a task = consecutive rows dealt to the chains,
// cut at `target_batches` batches. Offsets into B / C are pre-multiplied 32-bit byte offsets.
constexpr int kRecordPiece = 8;
constexpr int kRecordMaxRow = 1024;                // longer rows would pad the other chains of their task for too long
constexpr int64_t kRecordMaxBytes = 1ll << 34;     // of batches
struct RecordTables {
    void* block = nullptr;     // tasks + batches
    void* side = nullptr;      // slot + first (kept for gespmm_plan_set_values)
    int32_t* tasks = nullptr;  // ntasks int2 {first batch, #batches}
    char* batches = nullptr;
    int32_t* slot = nullptr;   // per row: first piece position inside its task * 16 + chain
    int32_t* row_task = nullptr;  // per row: its task
    int32_t* first = nullptr;  // per task: first batch (ntasks + 1)
    int32_t ntasks = 0;
    int32_t nbatches = 0;
    int32_t target_batches = 0;  // T: batches a task is cut at (longer when a row needs more)
    int32_t group = 0;         // W
    int64_t N = 0;
};
struct RecordArgs {
    const int32_t* tasks;
    const char* batches;
    const float* B;
    float* C;
    int32_t ntasks;
    int32_t n;
    const int32_t* guard;  // launch guard, as in SpmmArgs (NULL: no check)
    int32_t guard_want;
    // column-slab tables (plan.cpp: build_slab_tables; spmm_staged.hip only): the launch covers blocks blk0 .. blk0 + nblocks - 1 of the
    // tables; acc != 0: rows continue from the partial sums in C (task word 0 = C row of the task's first row, row-end codes carry the next)
    int32_t blk0;
    int32_t acc;
};
int records_group(int64_t N);  // lanes per chain at width N (0: width not served)
bool records_serves(int64_t M, int64_t K, int64_t N, int32_t max_degree);
// rowptr / colind / val: the matrix in the order its rows are processed (val NULL: 1.0f); perm: C row of row i (NULL: i)
hipError_t device_build_records(int64_t M, const int32_t* rowptr, const int32_t* colind, const float* val, const int32_t* perm,
                                int target_batches, int64_t N, RecordTables* out, hipStream_t st);
hipError_t device_records_set_values(const RecordTables& t, int64_t M, const int32_t* rowptr, const int32_t* colind, const float* val,
                                     hipStream_t st);
void free_records(RecordTables* t);
hipError_t launch_spmm_records(const RecordTables& t, const float* B, float* C, int64_t N, int flags, const LaunchGuard* guard, hipStream_t st);

// sddmm_kernels.hip
constexpr int kSddmmNoSlab = 1;  // launch_sddmm flag: never take the cache-blocked CSR form
hipError_t launch_sddmm(const int32_t* rowind_or_rowptr, bool csr, const int32_t* colind,
                        const float* D1, const float* D2, float* out,
                        int64_t M, int64_t nnz, int64_t N, int flags, hipStream_t st);
// split[(nslab+1)][M]: per-row forward-scan split points of the column slabs (spmm_kernels.hip)
hipError_t launch_slabplan(const int32_t* rowptr, const int32_t* colind, int32_t* split, int M, int nslab,
                           int slab_rows, hipStream_t st);

// baseline_kernels.hip — Gunrock-style edge-parallel atomicAdd scatter (comparison column only)
hipError_t launch_atomic_scatter(const int32_t* rowptr, const int32_t* colind, const float* in, float* out, int64_t M,
                                 int64_t K, int64_t N, int64_t nnz, hipStream_t st);

// ... and a plain streaming copy dst[i] = src[i] (the read + write rate of the box: bench.py's yardstick for ceiling_frac)
hipError_t launch_copy(const float* src, float* dst, int64_t n, hipStream_t st);

// csr2csc.hip
int64_t csr2csc_workspace_bytes(int64_t M, int64_t K, int64_t nnz);
hipError_t launch_csr2csc(const int32_t* rowptr, const int32_t* colind, const float* csr_val,
                          int32_t* colptr, int32_t* rowind, float* csc_val,
                          int64_t M, int64_t K, int64_t nnz, void* workspace, hipStream_t st);

}  // namespace gespmm
