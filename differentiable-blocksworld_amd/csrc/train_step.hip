// The whole optimisation iteration of the training path behind one entry point (include/dbw_hip.h: dbw_train_step_*).
//
// Replaces one pass of src/trainer.py:137-147 (zero_grad -> model(images) -> total.backward() -> optimizer.step()) for the decoupled
// training render (src/model/dbw.py:198-223, 361-408; src/model/renderer.py:84-98) with MSE + parsimony + TV + overlap.  This file is
// host-side orchestration: which kernel goes on which of three streams, in which order, and how the streams wait for each other -- plus
// a few small kernels of its own (tiling of the target images, reduction of the loss values, the one-thread store / poll through which the
// streams wait).  The arithmetic lives in the kernels it launches, each of which is the kernel the operator-level path uses or a fusion of
// several of them out of the same device functions (step_kernels.h), so a step computes what dbw_amd/native_step.py computes through ~33
// launches -- tests/test_gpu_c_step.py holds the two (and every fuse mask in between) to each other.
//
// Schedule with every fuse bit set (M = the caller's stream: the critical chain; E, Rg = the library's two lowest-priority streams of the
// device; `*` = the launch whose first workgroup stores the word the other streams poll -- see dbw_step_desc.sync_events):
//   M:  [tile targets] -> prologue (opacities, vertices) -> scene set-up* -> bins (+ texture preparation) -> launch order* -> fg pass (+ env
//       layer + composite + MSE) -> [poll: bin layout] fg backward* [-> bin reduction] -> poll (E) -> blocks' projection backward +
//       texture preparation backward -> blocks' tail -> Adam (clears the zero arena)
//   E:  poll (fg pass) -> env backward -> projection backward -> ground pose backward -> poll (Rg) -> env texture preparation backward -> store
//   Rg: poll (prologue) -> [bin cursors, layout -> store] -> regularisers -> poll (textures) -> TV -> poll (fg pass) -> loss values -> store
//       -> copy to the host
// Every poll is enqueued BEHIND the launch that carries its store (the order of the statements of dbw_train_step_run): that is what makes
// polling safe under any mapping of streams to hardware queues.  The legacy arrangements -- operator-level kernels, the env pass as a pass
// of its own, events instead of words, one stream -- are the same function with fuse bits / options cleared, and are what the tests
// compare the default with.
// (backward_order 1: the env backward waits for the fg backward KERNEL; data parallel with the whole gradient buffer reduced, the blocks'
// texture gradient -- 83 % of its bytes -- is then final early enough to be reduced next to the env chain.)
#include "dbw_common.h"
#include "raster_bin.h"
#include "step_kernels.h"
#include "../../include/dbw_hip.h"

#include <math.h>
#include <mutex>
#include <new>
#include <stdlib.h>
#include <string.h>

using namespace dbw;

namespace {

// ---- the two kernels of this file ---------------------------------------------------------------------------------------------------
// (B, 3, H, W) -> the 8x8-tile planar layout [B][ceil(H/8)][ceil(W/8)][3][64] (include/dbw_hip.h: image_layout 1); pixels beyond the
// image are zero.  One thread per element of the tiled image: the writes are coalesced, the reads are 32 B row pieces.
__global__ __launch_bounds__(256) void tile_target_kernel(const float *__restrict__ img, int B, int C, int H, int W, float *__restrict__ out) {
    const int tx = (W + 7) >> 3, ty = (H + 7) >> 3;
    const long long total = (long long)B * ty * tx * C * 64;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        long long r = i >> 6;
        const int c = (int)(r % C); r /= C;
        const int tix = (int)(r % tx); r /= tx;
        const int tiy = (int)(r % ty);
        const int n = (int)(r / ty);
        const int y = tiy * 8 + (lane >> 3), x = tix * 8 + (lane & 7);
        out[i] = (y < H && x < W) ? img[(((long long)n * C + c) * H + y) * W + x] : 0.f;
    }
}

// vals (device: parsimony, tv, overlap accumulated by their kernels in slots 1..3; slot 0 = this kernel's running sum, slot 5 its ticket)
// + the per-tile sums of squared differences of the fg pass -> out5 = rgb, parsimony, tv, overlap, total (what compute_losses returns,
// dbw.py:361-408); the workgroup that finishes last writes them
constexpr int LOSS_BLOCKS = 32;
__global__ __launch_bounds__(256) void loss_finish_kernel(const float *__restrict__ part, long long nparts, float scale, float tv_value_scale, float *vals,
                                                          float *__restrict__ out5) {
    __shared__ float s_red[4];
    __shared__ int s_last;
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nparts; i += (long long)gridDim.x * 256) acc += part[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        if (t != 0.f) unsafeAtomicAdd(vals, t);
        __threadfence();
        s_last = atomicAdd((unsigned *)(vals + 5), 1u) == gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        __threadfence();
        const float rgb = atomicAdd(vals, 0.f) * scale;          // (through the L2: the other workgroups' adds)
        const float tvv = vals[2] * tv_value_scale;
        out5[0] = rgb; out5[1] = vals[1]; out5[2] = tvv; out5[3] = vals[3];
        out5[4] = ((rgb + vals[1]) + tvv) + vals[3];
    }
}

size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct SceneBuf { size_t fvc, first, num, c2o, nbr, code, cw, rws, rws_bytes; };

struct Layout {
    SceneBuf e, f;
    size_t p2f_e, bary_e, dists_e, img_e, p2f, bary, dists, part, g_fg, g_env;
    size_t env_maps, blk_maps, sig[3], g_sig[3];
    size_t alpha, alpha_full, keep, blk_verts, sq_local, losses, target;
    size_t records, cursor[2], layout[2], layout_uniform;
    size_t arena_begin, g_alpha_full, ovl_ws, vals, tickets, g_blk_maps, g_env_maps, g_maps_end, g_fa, g_fvc_f, g_blk_verts, g_fvc_e, g_env_verts, arena_end;
    size_t total;
    // derived sizes
    int Fe, Ff, Ve, Vf, tiles, bin_cap;
    size_t ce, cb_, n_env_maps, n_blk_maps;
};

bool tv_on(const dbw_step_desc &d) { return d.w_tv_bkg != 0.f || d.w_tv_blocks != 0.f || d.w_tv_ground != 0.f; }
bool bins_on(const dbw_step_desc &d) { return d.decim_blocks == 1 && d.n_bins > 0 && d.block_bin_base && d.block_bin_info; }

int texbin_capacity(int B, int H, int W, int K, int nbins) {
    // records per texture bin: room for max(K / 2, 2) fragments per pixel spread evenly over the bins (a soft K-layer render fills ~20 %
    // of its slots; what does not fit falls back to atomics), at most 16 GiB, a multiple of the cursors per bin (dbw_amd/ops.py: the same)
    long long cap = (long long)B * H * W * (K > 4 ? K : 4) / (2LL * nbins);
    if (cap < 256) cap = 256;
    const long long hi = (16LL << 30) / (32LL * nbins);
    if (cap > hi) cap = hi;
    return (int)((cap + DBW_BIN_SUBCURSORS - 1) / DBW_BIN_SUBCURSORS * DBW_BIN_SUBCURSORS);
}

void scene_layout(SceneBuf &s, size_t &o, int B, int F, int H, int W) {
    const size_t n = (size_t)B * 2 * F;
    s.fvc = o; o += al(n * 9 * 4);
    s.first = o; o += al((size_t)B * 4);
    s.num = o; o += al((size_t)B * 4);
    s.c2o = o; o += al(n * 4);
    s.nbr = o; o += al(n * 4);
    s.code = o; o += al(n * 4);
    s.cw = o; o += al(n * 8);
    s.rws_bytes = dbw_rasterize_workspace_bytes_binned((int64_t)n, B, H, W);
    s.rws = o; o += al(s.rws_bytes);
}

void make_layout(const dbw_step_desc &d, Layout &L) {
    memset(&L, 0, sizeof(L));
    const int B = d.max_views, K = d.faces_per_pixel;
    L.Fe = d.n_sky_faces + d.n_ground_faces; L.Ff = d.n_blocks * d.block_nf;
    L.Ve = d.n_sky_verts + d.n_ground_verts; L.Vf = d.n_blocks * d.block_nv;
    L.tiles = ((d.H + 7) / 8) * ((d.W + 7) / 8);
    const size_t bt = (size_t)B * L.tiles;
    L.ce = (size_t)(d.env_txt_size / d.decim_env) * (d.env_txt_size / d.decim_env) * 3;
    L.cb_ = (size_t)(d.txt_size / d.decim_blocks) * (d.txt_size / d.decim_blocks) * 3;
    L.n_env_maps = 2 * L.ce; L.n_blk_maps = (size_t)d.n_blocks * L.cb_;
    size_t o = 0;
    scene_layout(L.e, o, B, L.Fe, d.H, d.W);
    scene_layout(L.f, o, B, L.Ff, d.H, d.W);
    L.p2f_e = o; o += al(bt * 64 * 4);
    L.bary_e = o; o += al(bt * 64 * 3 * 4);
    L.dists_e = o; o += al(bt * 64 * 4);
    L.img_e = o; o += al(bt * 256 * 4);
    L.p2f = o; o += al(bt * K * 64 * 4);
    L.bary = o; o += al(bt * K * 64 * 8 * 4);
    L.dists = o; o += al(bt * K * 64 * 4);
    L.part = o; o += al(bt * 4);
    L.g_fg = o; o += al(bt * 256 * 4);
    L.g_env = o; o += al(bt * 256 * 4);
    L.env_maps = o; o += al(L.n_env_maps * 4);
    L.blk_maps = o; o += al(L.n_blk_maps * 4);
    const size_t full[3] = {(size_t)d.env_txt_size * d.env_txt_size * 3, (size_t)d.n_blocks * d.txt_size * d.txt_size * 3, (size_t)d.env_txt_size * d.env_txt_size * 3};
    const int dec[3] = {d.decim_env, d.decim_blocks, d.decim_env};
    for (int i = 0; i < 3; ++i) {
        if (dec[i] > 1) { L.sig[i] = o; o += al(full[i] * 4); }
        else L.sig[i] = i == 0 ? L.env_maps : (i == 1 ? L.blk_maps : L.env_maps + L.ce * 4);     // undecimated: the maps ARE the sigmoid
        if (tv_on(d)) { L.g_sig[i] = o; o += al(full[i] * 4); }
    }
    L.alpha = o; o += al((size_t)d.n_blocks * 4);
    L.alpha_full = o; o += al((size_t)d.n_blocks * 4);
    L.keep = o; o += al((size_t)d.n_blocks * 4);
    L.blk_verts = o; o += al((size_t)L.Vf * 12);
    L.sq_local = o; o += al((size_t)L.Vf * 36);
    L.losses = o; o += al(8 * 4);
    L.target = o; o += al(bt * 192 * 4);
    if (bins_on(d)) {
        L.bin_cap = texbin_capacity(B, d.H, d.W, K, d.n_bins);
        const size_t nsub = (size_t)d.n_bins * DBW_BIN_SUBCURSORS;
        L.records = o; o += al((size_t)d.n_bins * L.bin_cap * 32);
        for (int i = 0; i < 2; ++i) { L.cursor[i] = o; o += al(nsub * 4); }
        for (int i = 0; i < 2; ++i) { L.layout[i] = o; o += al(nsub * 8); }
        L.layout_uniform = o; o += al(nsub * 8);
    }
    L.arena_begin = o;
    L.g_alpha_full = o; o += al((size_t)d.n_blocks * 4);
    L.ovl_ws = o; o += al((size_t)d.n_blocks * 18 * 4);
    L.vals = o; o += al(8 * 4);
    L.tickets = o; o += al(4 * 4);
    L.g_blk_maps = o; o += al(L.n_blk_maps * 4);          // (the gradients of the prepared maps: ONE range, what data-parallel ranks sum)
    L.g_env_maps = o; o += al(L.n_env_maps * 4);
    L.g_maps_end = o;
    L.g_fa = o; o += al((size_t)d.n_blocks * 64 * 4);
    L.g_fvc_f = o; o += al((size_t)B * 2 * L.Ff * 9 * 4);
    L.g_blk_verts = o; o += al((size_t)L.Vf * 12);
    L.g_fvc_e = o; o += al((size_t)B * 2 * L.Fe * 9 * 4);
    L.g_env_verts = o; o += al((size_t)L.Ve * 12);
    L.arena_end = o;
    L.total = o;
}


// the three texture tensors of the scene as the `_sets` kernels take them: sky, blocks, ground (dbw.py:273-293,306,331-334)
void fill_texture_sets(const dbw_step_desc &d, const Layout &L, char *ws, dbw_texture_set (&sets)[3]) {
#define FPW(off) ((float *)(ws + (off)))
    memset(sets, 0, sizeof(sets));
    const float *tex[3] = {d.texture_bkg, d.textures, d.texture_ground};
    float *gtex[3] = {d.g_texture_bkg, d.g_textures, d.g_texture_ground};
    const int tn[3] = {1, d.n_blocks, 1}, th[3] = {d.env_txt_size, d.txt_size, d.env_txt_size}, td[3] = {d.decim_env, d.decim_blocks, d.decim_env};
    float *maps_out[3] = {FPW(L.env_maps), FPW(L.blk_maps), FPW(L.env_maps) + L.ce};
    float *gmaps[3] = {FPW(L.g_env_maps), FPW(L.g_blk_maps), FPW(L.g_env_maps) + L.ce};
    const float tvw[3] = {d.w_tv_bkg, d.w_tv_blocks, d.w_tv_ground};
    for (int i = 0; i < 3; ++i) {
        dbw_texture_set &t = sets[i];
        t.texture = tex[i]; t.n = tn[i]; t.h = th[i]; t.w = th[i]; t.decim = td[i];
        t.maps = maps_out[i];
        t.sig = td[i] > 1 ? FPW(L.sig[i]) : nullptr;
        t.wrap_x = i == 1 ? 1 : 0;
        t.tv_scale = tvw[i];
        t.grad_texture = gtex[i];
        t.grad_maps = gmaps[i];
    }
}
// ... with the total-variation term on: where its gradient (to the sigmoid of the texture) goes, and comes from in the backward of the preparation
void add_tv_fields(const Layout &L, char *ws, dbw_texture_set (&sets)[3]) {
    for (int i = 0; i < 3; ++i) {
        sets[i].sig = FPW(L.sig[i]);
        sets[i].grad_sig_out = FPW(L.g_sig[i]);
        sets[i].grad_sig = FPW(L.g_sig[i]);
    }
#undef FPW
}

int check_desc(const dbw_step_desc *d) {
    DBW_REQUIRE(d, "null descriptor");
    DBW_REQUIRE(d->H > 0 && d->W > 0 && d->max_views > 0 && d->faces_per_pixel > 1 && d->faces_per_pixel <= DBW_MAX_FACES_PER_PIXEL, "bad image size / batch / faces_per_pixel (the soft pass has K > 1)");
    DBW_REQUIRE(d->n_blocks > 0 && d->n_blocks <= 64 && d->block_nv > 0 && d->block_nf > 0, "1..64 blocks");
    DBW_REQUIRE(d->n_blocks * d->block_nf < (1 << 20) && d->n_blocks + 2 < (1 << 11), "uv-fragments pack the face in 20 bits, the map in 11");
    DBW_REQUIRE(d->n_sky_verts > 0 && d->n_ground_verts > 0 && d->n_sky_faces > 0 && d->n_ground_faces > 0, "bad env mesh");
    DBW_REQUIRE(d->txt_size > 1 && d->env_txt_size > 1 && d->decim_env >= 1 && d->decim_blocks >= 1 && d->txt_size % d->decim_blocks == 0 &&
                    d->env_txt_size % d->decim_env == 0, "texture sizes must be multiples of their decimation factor");
    DBW_REQUIRE(d->sigma > 0.f && d->blur_radius >= 0.f && d->cam_eps > 0.f, "bad renderer constants");
    DBW_REQUIRE(d->R_world && d->T_world && d->Kmat && d->ground_base && d->env_verts && d->env_faces && d->env_face_uvs && d->env_face_map && d->env_map_desc &&
                    d->trig && d->block_faces && d->block_face_uvs && d->block_face_map && d->block_map_desc, "null table");
    DBW_REQUIRE(d->sq_eps && d->S && d->R6 && d->T && d->alpha_logit && d->R6_ground && d->T_ground && d->texture_bkg && d->texture_ground && d->textures, "null parameter");
    DBW_REQUIRE(d->g_sq_eps && d->g_S && d->g_R6 && d->g_T && d->g_alpha_logit && d->g_R6_ground && d->g_T_ground && d->g_texture_bkg && d->g_texture_ground &&
                    d->g_textures, "null gradient");
    DBW_REQUIRE(d->flat_param && d->flat_grad && d->exp_avg && d->exp_avg_sq && d->group_end[0] >= 0 && d->group_end[1] >= d->group_end[0], "bad Adam buffers");
    DBW_REQUIRE(d->small_grads && d->n_small_grads > 0, "small_grads: the accumulated gradients to clear");
    DBW_REQUIRE(d->w_rgb > 0.f && d->w_parsimony >= 0.f && d->w_overlap >= 0.f && d->w_tv_bkg >= 0.f && d->w_tv_blocks >= 0.f && d->w_tv_ground >= 0.f, "bad loss weights");
    DBW_REQUIRE(d->w_overlap == 0.f || (d->overlap_points > 0 && d->overlap_temperature > 0.f), "bad overlap constants");
    DBW_REQUIRE(((size_t)(d->n_sky_verts + d->n_ground_verts) * 12 <= 48 * 1024 && (size_t)d->n_blocks * d->block_nv * 12 <= 48 * 1024) || !(d->fuse & 8),
                "fused tails keep a scene's vertex gradients in 48 KB of LDS");
    // (the fused forward addresses its texels with 32-bit byte offsets from the first prepared map of a scene: include/dbw_hip.h, frag_layout 2)
    DBW_REQUIRE((size_t)d->n_blocks * d->txt_size * d->txt_size * 3 < ((size_t)1 << 30) &&
                    (size_t)2 * d->env_txt_size * d->env_txt_size * 3 < ((size_t)1 << 30), "a scene's prepared maps must hold fewer than 2^30 floats");
    return DBW_OK;
}

// The two side streams of the training step: ONE pair per process and device, shared by every plan, created on first use and never
// destroyed.  HIP multiplexes a process's streams onto a handful of hardware queues round-robin: when every plan made its own pair, the
// second plan of a process (the next training phase) got streams that share a queue with the caller's stream -- false dependencies
// between "concurrent" chains, 1.6 -> 2.2 ms per step at the full-resolution phase.  Lowest priority: whatever shares the GPU with the
// critical chain on the caller's stream yields to it.
constexpr int MAX_DEVICES = 64;
hipStream_t g_stream_r[MAX_DEVICES], g_stream_env[MAX_DEVICES];
std::mutex g_stream_lock;
int step_streams(hipStream_t &r, hipStream_t &e) {
    std::lock_guard<std::mutex> hold(g_stream_lock);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) { dbw_set_error("dbw_train_step_create: no current device"); return DBW_ERR_LAUNCH; }
    if (!g_stream_r[dev]) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&g_stream_r[dev], hipStreamNonBlocking, least) != hipSuccess ||
            hipStreamCreateWithPriority(&g_stream_env[dev], hipStreamNonBlocking, least) != hipSuccess) {
            dbw_set_error("dbw_train_step_create: hipStreamCreate failed");
            return DBW_ERR_LAUNCH;
        }
    }
    r = g_stream_r[dev]; e = g_stream_env[dev];
    return DBW_OK;
}

// Cross-stream ordering through memory (dbw_step_desc.sync_events == 0): the producer's stream stores a counter behind its work, the
// consumer's stream polls it in front of its own.  Kernel boundaries do the rest: the producer's kernels have released their writes before
// the store kernel starts, and the kernel behind the poll acquires at its start like any kernel behind an event wait.
constexpr int SYNC_FLAGS = 12, SYNC_TIMEOUT_SLOT = 15, SYNC_WORDS = 32;      // (words 16 .. 27: the counters as the first poll that gave up saw them)
constexpr int SYNC_VOID_SLOT = 28;       // the plan's "a poll gave up" word (float 1.0), see sync_wait_kernel
enum { F_PROLOGUE, F_SCATTER, F_FG_FWD, F_REG, F_LAYOUT, F_KERNEL_DONE, F_BLOCKS_READY, F_ENV_DONE, F_TEX };
__global__ void sync_set_kernel(unsigned *flag, unsigned v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
// (timeouts: a counter in device memory; host_timeouts: the same in mapped host memory -- the next dbw_train_step_run sees it without a
// transfer and fails loudly: a poll that gave up let its stream run ahead of what it was waiting for)
// A poll that gives up (`limit` ticks of the 100 MHz wall clock: 1 s; the order of the enqueues rules a deadlock out, so this only ever
// fires when the process's queues were descheduled that long) VOIDS the step instead of letting it update anything: it raises the plan's
// `void_raised` word, counts itself in device memory, and leaves a nonzero word in mapped host memory (a plain system-scope store: no PCIe
// atomics needed) that the next dbw_train_step_run sees without a transfer: from then on the plan orders its streams through events.
// `void_raised` is STICKY: nothing on the device ever clears it -- a clear at the head of a run, ordered on the main stream, is exactly
// what a main stream that was stalled for a second behind the caller's earlier work would execute AFTER its side streams' polls had
// given up, wiping the flag in front of an Adam launch that then applies gradients of stale inputs.  Every run latches the word into the
// step's `void_flag` behind its join (void_latch: the tail kernel's first thread) -- the float the run's Adam launch, and summed over
// the ranks every rank's, reads: != 0 -> no parameter, no moment moves, the arena is still cleared -- so every run enqueued between the
// poll that gave up and the host noticing is voided too; the host clears the word once it has synchronised the device (the head of the
// next dbw_train_step_run) and the plan goes on through events.
__global__ void void_latch_kernel(const float *void_raised, float *void_flag) { *void_flag = *void_raised; }
__global__ void sync_wait_kernel(const unsigned *flag, unsigned v, unsigned *timeouts, unsigned *host_timeouts, float *void_flag, unsigned long long limit) {
    const unsigned long long t0 = wall_clock64();            // 100 MHz
    // (a poll only gives up after it has itself been RUNNING for a good part of the limit -- `spins`, about a microsecond each: wall-clock
    // time that passed while the process's queues were off the hardware, e.g. while the driver clears a freshly allocated workspace, is
    // nobody's failure to signal)
    const unsigned long long min_spins = limit >> 8;
    unsigned long long spins = 0;
    while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - v) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > min_spins && wall_clock64() - t0 > limit) {
            atomicAdd(timeouts, 1u);
            // (diagnostics, dbw_debug_train_step_last_timeout: which counter the first poll that gave up was waiting on, the value it wanted and
            // the value it last saw -- the three words in front of the count)
            if (atomicCAS(timeouts - 3, 0u, (unsigned)(flag - (timeouts - SYNC_TIMEOUT_SLOT)) + 1u) == 0u) {
                timeouts[-2] = v;
                timeouts[-1] = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                for (int i = 0; i < SYNC_FLAGS; ++i) timeouts[1 + i] = __hip_atomic_load(timeouts - SYNC_TIMEOUT_SLOT + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            }
            __hip_atomic_store(void_flag, 1.f, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(host_timeouts, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
}
constexpr unsigned long long SYNC_LIMIT_TICKS = 100000000ull;      // 1 s

}  // namespace

struct dbw_step_plan {
    dbw_step_desc d;
    Layout L;
    char *ws;
    RasterWorkspace rw_e, rw_f;         // for max_views (the pointers of a run follow from the run's own B)
    hipEvent_t ev_prologue, ev_scatter, ev_fg_fwd, ev_reg, ev_layout, ev_kernel_done, ev_blocks_ready, ev_env_done, ev_losses;
    hipStream_t stream_r, stream_env;   // the library's two side streams of this device (step_streams): the regularisers; the env backward chain
    unsigned long long runs;            // completed calls of dbw_train_step_run
    int bin_turn, bin_ready, uniform_ready;
    bool arena_clean;
    float *host_losses;                 // pinned
    bool losses_pending;
    bool phase1_done;                   // a phase-1 run is waiting for its phase 2
    bool cur_flags;                     // the step in progress (or the last one) orders its streams through polled words, not events
    bool profile, profiled;             // dbw_train_step_profile: timing events around the four big kernels of a run
    hipEvent_t ev_t[8];
    unsigned *sync_words;               // device: SYNC_FLAGS counters + the number of polls that gave up
    unsigned *host_timeouts, *host_timeouts_dev;      // ... and the same number in mapped host memory (host pointer, device pointer)
    unsigned sync_val[SYNC_FLAGS];      // last value stored behind each counter (host side)
    int voided_runs;                    // runs whose cross-stream wait gave up (seen at the next run): the plan then runs on events
    int force_timeout;                  // debug (dbw_debug_train_step_force_timeout): the next run's join polls for a value that never comes
};

extern "C" size_t dbw_train_step_workspace_bytes(const dbw_step_desc *desc) {
    if (check_desc(desc)) return 0;
    Layout L;
    make_layout(*desc, L);
    return L.total;
}

extern "C" dbw_step_plan *dbw_train_step_create(const dbw_step_desc *desc, void *workspace, size_t workspace_bytes) {
    if (check_desc(desc)) return nullptr;
    if (!workspace || ((uintptr_t)workspace & 255)) { dbw_set_error("dbw_train_step_create: the workspace must be 256-byte aligned"); return nullptr; }
    dbw_step_plan *p = new (std::nothrow) dbw_step_plan();        // (value-initialised: every handle below starts out null, and
    if (!p) { dbw_set_error("dbw_train_step_create: out of host memory"); return nullptr; }      // dbw_train_step_destroy skips what is null)
    p->d = *desc;
    make_layout(p->d, p->L);
    if (workspace_bytes < p->L.total) {
        dbw_set_error("dbw_train_step_create: workspace of %zu bytes, %zu needed", workspace_bytes, p->L.total);
        dbw_train_step_destroy(p);
        return nullptr;
    }
    p->ws = (char *)workspace;
    hipEvent_t *evs[] = {&p->ev_prologue, &p->ev_scatter, &p->ev_fg_fwd, &p->ev_reg, &p->ev_layout, &p->ev_kernel_done, &p->ev_blocks_ready, &p->ev_env_done, &p->ev_losses};
    for (hipEvent_t *e : evs)
        if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) { dbw_set_error("dbw_train_step_create: hipEventCreate failed"); dbw_train_step_destroy(p); return nullptr; }
    if (step_streams(p->stream_r, p->stream_env)) { dbw_train_step_destroy(p); return nullptr; }
    if (hipHostMalloc((void **)&p->host_losses, 8 * sizeof(float), hipHostMallocDefault) != hipSuccess) {
        dbw_set_error("dbw_train_step_create: hipHostMalloc failed");
        dbw_train_step_destroy(p);
        return nullptr;
    }
    for (int i = 0; i < 8; ++i) p->host_losses[i] = 0.f;
    p->sync_words = nullptr;
    if (hipMalloc((void **)&p->sync_words, SYNC_WORDS * sizeof(unsigned)) != hipSuccess || hipMemset(p->sync_words, 0, SYNC_WORDS * sizeof(unsigned)) != hipSuccess) {
        dbw_set_error("dbw_train_step_create: hipMalloc failed");
        dbw_train_step_destroy(p);
        return nullptr;
    }
    for (unsigned &v : p->sync_val) v = 0;
    p->host_timeouts = nullptr; p->host_timeouts_dev = nullptr;
    if (hipHostMalloc((void **)&p->host_timeouts, 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&p->host_timeouts_dev, p->host_timeouts, 0) != hipSuccess) {
        dbw_set_error("dbw_train_step_create: hipHostMalloc (mapped) failed");
        dbw_train_step_destroy(p);
        return nullptr;
    }
    *p->host_timeouts = 0u;
    p->profile = p->profiled = false;
    p->phase1_done = false;
    for (hipEvent_t &e : p->ev_t)
        if (hipEventCreate(&e) != hipSuccess) { dbw_set_error("dbw_train_step_create: hipEventCreate failed"); dbw_train_step_destroy(p); return nullptr; }
    p->runs = 0; p->cur_flags = false; p->voided_runs = 0; p->force_timeout = 0; p->bin_turn = 0; p->bin_ready = 0; p->uniform_ready = 0; p->arena_clean = false; p->losses_pending = false;
    return p;
}

extern "C" void dbw_train_step_destroy(dbw_step_plan *p) {
    if (!p) return;
    hipEvent_t evs[] = {p->ev_prologue, p->ev_scatter, p->ev_fg_fwd, p->ev_reg, p->ev_layout, p->ev_kernel_done, p->ev_blocks_ready, p->ev_env_done, p->ev_losses};
    for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->ev_t) if (e) (void)hipEventDestroy(e);
    if (p->host_losses) (void)hipHostFree(p->host_losses);
    if (p->sync_words) (void)hipFree(p->sync_words);
    if (p->host_timeouts) (void)hipHostFree(p->host_timeouts);
    delete p;
}

extern "C" int64_t dbw_train_step_offset(const dbw_step_plan *p, int which) {
    if (!p) return -1;
    const Layout &L = p->L;
    switch (which) {
        case 0: return (int64_t)L.alpha;
        case 1: return (int64_t)L.alpha_full;
        case 2: return (int64_t)L.keep;
        case 3: return (int64_t)L.losses;
        case 4: return (int64_t)L.arena_begin;
        case 5: return (int64_t)L.arena_end;
        case 6: return (int64_t)L.g_fg;
        case 7: return (int64_t)L.g_env;
        case 8: return (int64_t)L.img_e;
        case 9: return (int64_t)L.blk_verts;
        case 10: return (int64_t)L.part;
        case 11: return (int64_t)L.p2f_e;
        case 12: return (int64_t)L.bary_e;
        case 13: return (int64_t)L.g_blk_maps;
        case 14: return (int64_t)L.g_maps_end;
        default: return -1;
    }
}


extern "C" int dbw_train_step_losses(dbw_step_plan *p, float *out5) {
    DBW_REQUIRE(p && out5, "null pointer");
    DBW_REQUIRE(p->losses_pending, "no run with read_losses != 0 to read from");
    if (hipEventSynchronize(p->ev_losses) != hipSuccess) { dbw_set_error("dbw_train_step_losses: hipEventSynchronize failed"); return DBW_ERR_LAUNCH; }
    for (int i = 0; i < 5; ++i) out5[i] = p->host_losses[i];
    return DBW_OK;
}

#define HIP_OK(call)                                                              \
    do {                                                                          \
        const hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) {                                                   \
            dbw_set_error("dbw_train_step_run: %s: %s", #call, hipGetErrorString(e_)); \
            return DBW_ERR_LAUNCH;                                                \
        }                                                                         \
    } while (0)
#define RC(call)                  \
    do {                          \
        const int rc_ = (call);   \
        if (rc_) return rc_;      \
    } while (0)

extern "C" int dbw_train_step_run(dbw_step_plan *p, const dbw_step_inputs *in, dbw_stream_t stream_main, dbw_stream_t stream_side) {
    DBW_REQUIRE(p && in, "null pointer");
    const dbw_step_desc &d = p->d;
    const Layout &L = p->L;
    DBW_REQUIRE(in->imgs && in->R && in->T, "null input");
    if (*(volatile unsigned *)p->host_timeouts != 0u) {
        // A cross-stream poll of the previous run gave up: that run voided itself on the device (void_flag: no parameter moved, its arena was
        // cleared; data-parallel ranks saw the flag through their gradient sum and skipped the update too).  From here on this plan orders
        // its streams through events -- the form that cannot give up -- and the caller simply goes on: one optimisation step was lost.
        HIP_OK(hipDeviceSynchronize());
        *(volatile unsigned *)p->host_timeouts = 0u;
        HIP_OK(hipMemset(p->sync_words + SYNC_VOID_SLOT, 0, sizeof(unsigned)));       // (the sticky word: cleared by nobody but the host, here)
        p->d.sync_events = 1;
        p->voided_runs += 1;
        p->arena_clean = false;          // (whatever the voided run left in the arena: this run opens with a fill)
    }
    DBW_REQUIRE(in->B >= 1 && in->B <= d.max_views, "B must lie in [1, max_views]");
    DBW_REQUIRE(in->global_count > 0.0, "global_count must be positive");
    DBW_REQUIRE(!in->with_adam || in->adam_step >= 1, "adam_step >= 1");
    const bool defer = in->defer_textures != 0;       // stop in front of the texture preparation's backward (dbw_train_step_finish runs it)
    DBW_REQUIRE(!defer || !in->with_adam, "defer_textures: Adam runs in dbw_train_step_finish");
    const int phase = in->phase;          // 0: the whole iteration; 1: up to the fg pass, which stores rec; 2: the fg pass again with grad_rec, then the rest
    DBW_REQUIRE(phase >= 0 && phase <= 2 && (phase != 1 || in->rec_out) && (phase != 2 || in->grad_rec), "phase 1 stores rec_out, phase 2 reads grad_rec");
    DBW_REQUIRE(phase == 0 || ((p->d.fuse & 18) == 18), "the two-phase iteration (perceptual term) needs the env layer folded into the fg pass (fuse bits 1 and 4)");
    DBW_REQUIRE(phase != 2 || p->phase1_done, "phase 2 without a phase 1 in front of it");
    const bool head = phase != 2, rest = phase != 1;
    // M: the critical chain.  E: the env backward chain, Rg: the regularisers -- the caller's side stream, or (NULL) streams of the plan
    // at the lowest priority, so that whatever shares the GPU with the fg chain yields to it.  stream_side == stream_main: one stream.
    hipStream_t M = (hipStream_t)stream_main;
    const bool two = !in->single_stream;
    hipStream_t E = !two ? M : (stream_side ? (hipStream_t)stream_side : p->stream_env), Rg = two ? p->stream_r : M;
    // one stream waiting for another: memory flags (default) or events, see dbw_step_desc.sync_events.  Every wait is enqueued AFTER the
    // signal it waits for -- the order of the statements below -- which is what makes the polling form safe
    // The FIRST run of a plan always goes through events: it is the run that pays for everything lazy -- measured at BASELINE config 5: the
    // first kernel that touches the freshly allocated 40 GB of workspace and fragment buffers starts 0.96 s late (the driver clears new
    // video memory behind the allocation), while the side streams' polls, which touch none of it, are already running against their
    // one-second budget (tools/diag/c5_after.py).  A two-phase step keeps the form its first phase chose.
    if (head) p->cur_flags = d.sync_events == 0 && p->runs > 0;
    const bool flags = p->cur_flags;
    auto signal = [&](hipStream_t st, int idx, hipEvent_t ev) -> int {
        if (!flags) { HIP_OK(hipEventRecord(ev, st)); return DBW_OK; }
        hipLaunchKernelGGL(sync_set_kernel, dim3(1), dim3(1), 0, st, p->sync_words + idx, ++p->sync_val[idx]);
        return dbw_check_launch("sync_set_kernel");
    };
    char *ws = p->ws;
    float *void_flag = (float *)(ws + L.losses) + 7;          // this run's copy of void_raised, latched behind the join; read by Adam
    float *void_raised = (float *)(p->sync_words + SYNC_VOID_SLOT);      // raised by a poll that gave up; sticky until the host has seen it
    const int force_timeout = p->force_timeout;
    p->force_timeout = 0;
    auto await = [&](hipStream_t st, int idx, hipEvent_t ev) -> int {
        if (!flags) { HIP_OK(hipStreamWaitEvent(st, ev, 0)); return DBW_OK; }
        const bool forced = force_timeout == 1 && idx == F_ENV_DONE;       // (tests: a value that never comes, a 0.05 s limit)
        const bool hasty = force_timeout == 2 && idx == F_PROLOGUE;        // (tests: the real value, a 0.02 s limit -- in front of a stalled main stream)
        hipLaunchKernelGGL(sync_wait_kernel, dim3(1), dim3(1), 0, st, (const unsigned *)(p->sync_words + idx), p->sync_val[idx] + (forced ? 0x10000000u : 0u),
                           p->sync_words + SYNC_TIMEOUT_SLOT, p->host_timeouts_dev, void_raised, forced ? 5000000ull : hasty ? 2000000ull : SYNC_LIMIT_TICKS);
        return dbw_check_launch("sync_wait_kernel");
    };
#define FP(off) ((float *)(ws + (off)))
#define IP(off) ((int *)(ws + (off)))
    const int B = in->B, H = d.H, W = d.W, K = d.faces_per_pixel, nb = d.n_blocks, nv = d.block_nv;
    const int Fe = L.Fe, Ff = L.Ff, Ve = L.Ve, Vf = L.Vf;
    const bool coarse = d.coarse != 0, bins = bins_on(d), tv = tv_on(d);
    const float mse_scale = (float)((double)d.w_rgb / in->global_count);
    const int zc_on = d.z_clip > 0.f ? 1 : 0;
    const bool noise_on = coarse && d.opacity_noise != 0.f;
    const float *fa = coarse ? FP(L.alpha) : nullptr;
    const int alpha_len = coarse ? -nb : 0;
    const int64_t Fte = (int64_t)B * 2 * Fe, Ftf = (int64_t)B * 2 * Ff;
    const bool overlap_on = d.w_overlap != 0.f, pars_on = d.w_parsimony != 0.f;
    DBW_REQUIRE((d.fuse & 1) || !noise_on || in->noise_override, "the operator-level prologue needs the caller's opacity noise (noise_override)");
    DBW_REQUIRE((d.fuse & 4) || !overlap_on || in->overlap_u_override, "the operator-level regularisers need the caller's overlap samples (overlap_u_override)");

    // the zero arena: cleared by the plan's own Adam launch at the end of a run; before the first run (and after a run whose caller ran
    // Adam itself without clearing it) by a fill
    if (head && !p->arena_clean && !(in->arena_is_clean && p->runs > 0)) HIP_OK(hipMemsetAsync(ws + L.arena_begin, 0, L.arena_end - L.arena_begin, M));
    p->arena_clean = false;

    // ---- texture sets: sky, blocks, ground (dbw.py:273-293,306,331-334) ----
    dbw_texture_set sets[3];
    fill_texture_sets(d, L, ws, sets);

    // ---- M: targets in the tile-planar layout (a fresh mini-batch; resident views come tiled) ----
    const float *target = in->imgs;
    if (!in->imgs_tiled) {
        const long long total = (long long)B * L.tiles * 192;
        long long g = (total + 255) / 256;
        if (g > 4096) g = 4096;
        if (head) {
            hipLaunchKernelGGL(tile_target_kernel, dim3((unsigned)g), dim3(256), 0, M, in->imgs, B, 3, H, W, FP(L.target));
            RC(dbw_check_launch("tile_target_kernel"));
        }
        target = FP(L.target);
    }

    // ---- M: camera transform, clipping, per-face records, bins of both scenes, launch order of the fg pass's tiles ----
    RasterWorkspace we, wf;
    // (fuse bit 4: the env layer is evaluated inside the fg pass, from per-tile lists of its own; else the hard pass walks its coarse bins)
    RC(dbw_raster_workspace_layout(ws + L.e.rws, L.e.rws_bytes, Fte, 2 * Fe, B, H, W, (d.fuse & 16) != 0, we));
    RC(dbw_raster_workspace_layout(ws + L.f.rws, L.f.rws_bytes, Ftf, 2 * Ff, B, H, W, true, wf));
    const bool fused_setup = (d.fuse & 2) && we.binned && wf.binned && wf.cells;
    const bool fold = fused_setup && (d.fuse & 16) && we.cells;
    const float margin_f = (float)sqrt((double)d.blur_radius);
    // large batches: the blocks' set-up runs on E next to the env pass (which is long enough to hide the hop); small ones: on M
    const bool setup_aside = two && fused_setup && !fold && B > d.serial_setup_max_views;
    DBW_REQUIRE(phase == 0 || fold, "the two-phase iteration needs per-tile lists for both scenes (a binned workspace)");
    // the step's texture preparation in the shadow of the bins (fuse bit 6): the prologue then only computes what the set-up waits for.  Its
    // consumers -- the fg pass (behind the bins on M) and the TV term on Rg, which polls a word the launch behind the bins stores (memory
    // words only: an event recorded there would cost M what this saves)
    const bool tex_in_bins = (d.fuse & 64) && (d.fuse & 1) && flags && two && fused_setup && !setup_aside;
    // ---- M: prologue ----
    const float thresh = d.mask_threshold;
    if (!head) {
    } else if (d.fuse & 1) {
        PrologueArgs P;
        memset(&P, 0, sizeof(P));
        for (int i = 0; i < 3; ++i) P.tex.s[i] = sets[i];
        P.tex.s[3] = sets[0];
        P.nsets = tex_in_bins ? 0 : 3;
        P.alpha_logit = d.alpha_logit; P.noise = noise_on ? in->noise_override : nullptr;
        P.noise_scale = noise_on ? d.opacity_noise : 0.f; P.thresh = thresh; P.nb = nb;
        P.alpha = FP(L.alpha); P.alpha_full = FP(L.alpha_full); P.keep = IP(L.keep);
        P.seed = d.seed; P.rng_step = in->rng_step;
        P.sq_eps = d.sq_eps; P.S = d.S; P.R6 = d.R6; P.T = d.T; P.trig = d.trig; P.nv = nv;
        P.ratio = d.ratio_block_scene; P.scale_min = d.scale_min; P.S_world = d.S_world; P.Rw = d.R_world; P.Tw = d.T_world;
        P.blk_verts = FP(L.blk_verts);
        P.sq_local = FP(L.sq_local);
        P.ground_base = d.ground_base; P.ngv = d.n_ground_verts; P.R6g = d.R6_ground; P.Tg = d.T_ground;
        P.ground_verts = d.env_verts + (size_t)d.n_sky_verts * 3;
        P.zero0 = d.small_grads; P.nzero0 = d.n_small_grads;
        // The two-phase iteration (a network of the caller's -- LPIPS-VGG16: ~330 MIOpen / rocBLAS / torch launches, 12 ms -- runs between the
        // phases on this stream): behind it, THESE 636 bytes of stores kept the prologue busy for 1.8 ms, step after step (2.1 of the 2.5 ms the
        // render path cost such a step; bisected down to this loop, profiles/r05_experiments.md: not the polls, not events, not scratch, not
        // clocks, not the host; a fill node in front of the launch writes the same bytes in 2 us).  So there the fill node does it.
        if (phase == 1) { HIP_OK(hipMemsetAsync(d.small_grads, 0, (size_t)d.n_small_grads * 4, M)); P.nzero0 = 0; }
        RC(launch_step_prologue(P, M));
    } else {
        RC(dbw_posed_mesh_fwd(d.ground_base, d.n_ground_verts, d.R6_ground, d.T_ground, d.S_world, d.R_world, d.T_world,
                              d.env_verts + (size_t)d.n_sky_verts * 3, M));
        RC(dbw_texture_prep_fwd_sets(sets, 3, M));
        HIP_OK(hipMemsetAsync(d.small_grads, 0, (size_t)d.n_small_grads * 4, M));
        RC(dbw_block_alpha_fwd(d.alpha_logit, noise_on ? in->noise_override : nullptr, noise_on ? d.opacity_noise : 0.f, thresh, nb, FP(L.alpha),
                               FP(L.alpha_full), IP(L.keep), M));
        RC(dbw_sq_blocks_fwd(d.sq_eps, d.S, d.R6, d.T, d.trig, IP(L.keep), 0, nb, nv, d.ratio_block_scene, d.scale_min, d.S_world, d.R_world,
                             d.T_world, FP(L.blk_verts), M));
    }
    // (measured: a chain of dependent kernels enqueued from here on ONE stream runs without gaps; an event costs the stream that records or
    // waits for it ~7 us before its next kernel, and a kernel behind an event of ANOTHER stream starts 12-26 us after that event.  So the
    // critical chain -- set-up, passes, fg backward, its tail, Adam -- stays on M and only what is off it forks)
    // The prologue's signal to Rg: carried by the first workgroup of the kernel behind the prologue on M (a kernel that has started says
    // that everything in front of it on its stream is complete) -- then Rg's work is enqueued BEHIND that kernel, because a poll must never
    // be enqueued in front of its producer -- or, where M's next kernel is not the fused set-up, by a launch of its own
    const bool prologue_signal_folded = flags && two && head && fused_setup && !setup_aside;
    if (two && head && !prologue_signal_folded) RC(signal(M, F_PROLOGUE, p->ev_prologue));
    int *cursor = nullptr;
    const uint32_t *blayout = nullptr;
    float *vals = FP(L.vals);
    auto side_head = [&]() -> int {
        if (two && head) RC(await(Rg, F_PROLOGUE, p->ev_prologue));

        // ---- Rg: texture bins: this step's cursors and record sub-ranges; the regularisers, value + gradient in one pass, weights folded into
        // the kernels' scales (dbw.py:373-405) ----
        if (bins) {
            const int64_t nsub = (int64_t)d.n_bins * DBW_BIN_SUBCURSORS;
            const double total_records = (double)d.n_bins * (double)L.bin_cap;
            cursor = IP(L.cursor[p->bin_turn]);
            if (head) HIP_OK(hipMemsetAsync(cursor, 0, (size_t)nsub * 4, Rg));
            if (p->bin_ready) {            // sub-ranges by the demand of the previous run (its cursors), ops.BinDemand
                if (head) RC(dbw_bin_layout(IP(L.cursor[1 - p->bin_turn]), nsub, total_records, 64, (uint32_t *)(ws + L.layout[p->bin_turn]), Rg));
                blayout = (const uint32_t *)(ws + L.layout[p->bin_turn]);
            } else {                       // first run: equal shares = the layout of an all-zero demand (this run's fresh cursors)
                if (head) RC(dbw_bin_layout(cursor, nsub, total_records, 1, (uint32_t *)(ws + L.layout_uniform), Rg));
                blayout = (const uint32_t *)(ws + L.layout_uniform);
            }
            if (two && head) RC(signal(Rg, F_LAYOUT, p->ev_layout));
        }
        if (!head) {
        } else if (d.fuse & 4) {
            RegulariserArgs A;
            memset(&A, 0, sizeof(A));
            A.u = overlap_on ? in->overlap_u_override : nullptr; A.npts = d.overlap_points; A.seed = d.seed; A.rng_step = in->rng_step;
            A.sq_eps = d.sq_eps; A.S = d.S; A.R6 = d.R6; A.T = d.T; A.alpha_full = FP(L.alpha_full); A.nb = nb;
            A.ratio = d.ratio_block_scene; A.scale_min = d.scale_min; A.inv_temp = overlap_on ? 1.f / d.overlap_temperature : 1.f; A.thresh = d.overlap_n_blocks;
            A.overlap_scale = d.w_overlap;
            A.pars_eps = 1e-6f; A.pars_scale = d.w_parsimony;
            A.loss_parsimony = vals + 1; A.loss_overlap = vals + 3;
            A.g_sq_eps = d.g_sq_eps; A.g_S = d.g_S; A.g_R6 = d.g_R6; A.g_T = d.g_T; A.g_alpha_full = FP(L.g_alpha_full);
            A.ws = FP(L.ovl_ws); A.ticket = (unsigned *)(ws + L.tickets);
            RC(launch_regularisers(A, Rg));
        } else {
            if (pars_on) RC(dbw_sqrt_mean(FP(L.alpha_full), nb, 1e-6f, d.w_parsimony, vals + 1, FP(L.g_alpha_full), Rg));
            if (overlap_on)
                RC(dbw_overlap_loss(in->overlap_u_override, d.overlap_points, d.sq_eps, d.S, d.R6, d.T, FP(L.alpha_full), nb, d.ratio_block_scene, d.scale_min,
                                    d.overlap_temperature, d.overlap_n_blocks, d.w_overlap, vals + 3, d.g_sq_eps, d.g_S, d.g_R6, d.g_T, FP(L.g_alpha_full),
                                    FP(L.ovl_ws), Rg));
        }
        if (tv) {
            add_tv_fields(L, ws, sets);
            if (head && tex_in_bins) RC(await(Rg, F_TEX, p->ev_prologue));     // (the sigmoid of the textures: written next to the bins)
            if (head) RC(dbw_tv_l2sq_sets(sets, 3, vals + 2, Rg));
        }
        return DBW_OK;
    };
    if (!prologue_signal_folded) RC(side_head());

    if (!head) {
    } else if (fused_setup) {
        SceneSetupArgs A;
        memset(&A, 0, sizeof(A));
        A.R = in->R; A.T = in->T; A.Kmat = d.Kmat; A.B = B;
        SceneGeom &e = A.sc[0], &f = A.sc[1];
        e.verts = d.env_verts; e.faces = d.env_faces; e.V = Ve; e.F = Fe;
        f.verts = FP(L.blk_verts); f.faces = d.block_faces; f.V = Vf; f.F = Ff;
        const SceneBuf *sb[2] = {&L.e, &L.f};
        RasterWorkspace *rw[2] = {&we, &wf};
        for (int i = 0; i < 2; ++i) {
            SceneGeom &g = A.sc[i];
            g.cam_eps = d.cam_eps; g.zc_on = zc_on; g.zc = d.z_clip; g.persp = d.perspective_correct;
            g.fvc = FP(sb[i]->fvc); g.first_idx = IP(sb[i]->first); g.num_faces = IP(sb[i]->num); g.c2o = IP(sb[i]->c2o); g.neighbor = IP(sb[i]->nbr);
            g.code = IP(sb[i]->code); g.cw = FP(sb[i]->cw);
            g.bbox = rw[i]->bbox; g.recs = rw[i]->recs;
        }
        e.margin = 0.f; f.margin = margin_f;
        f.hdr = wf.hdr; f.nhdr = CELL_HDR_INTS;
        f.srec = wf.shade_recs; f.face_uvs = d.block_face_uvs; f.face_map = d.block_face_map; f.map_desc = d.block_map_desc; f.map_alpha = fa;
        SceneBinsArgs Bn;
        memset(&Bn, 0, sizeof(Bn));
        Bn.B = B; Bn.H = H; Bn.W = W; Bn.nx = wf.nx; Bn.ny = wf.ny;
        for (int i = 0; i < 2; ++i) {
            SceneBinsArgs::One &g = Bn.sc[i];
            g.bbox = rw[i]->bbox; g.recs = rw[i]->recs; g.first_idx = IP(sb[i]->first); g.num_faces = IP(sb[i]->num);
            g.list = rw[i]->list; g.count = rw[i]->count; g.mask = rw[i]->mask;
        }
        Bn.sc[1].cells = 1; Bn.sc[1].cell = wf.cell; Bn.sc[1].pool = wf.pool; Bn.sc[1].pool_cap = wf.pool_cap; Bn.sc[1].hdr = wf.hdr; Bn.sc[1].rank = wf.rank;
        if (fold) {          // the env scene gets shading records and per-tile lists too
            e.hdr = we.hdr; e.nhdr = CELL_HDR_INTS;
            e.srec = we.shade_recs; e.face_uvs = d.env_face_uvs; e.face_map = d.env_face_map; e.map_desc = d.env_map_desc; e.map_alpha = nullptr;
            Bn.sc[0].cells = 1; Bn.sc[0].cell = we.cell; Bn.sc[0].pool = we.pool; Bn.sc[0].pool_cap = we.pool_cap; Bn.sc[0].hdr = we.hdr; Bn.sc[0].rank = we.rank;
            Bn.sc[0].dom = we.dom;       // ... and the dominant face of every tile (env_fold_pixel's fast path)
        }
        if (setup_aside) {
            RC(await(E, F_PROLOGUE, p->ev_prologue));
            A.scene0 = 0; A.nscenes = 1; Bn.scene0 = 0; Bn.nscenes = 1;
            RC(launch_scene_setup(A, M));
            RC(launch_scene_bins(Bn, M));
            A.scene0 = 1; Bn.scene0 = 1;
            RC(launch_scene_setup(A, E));
            RC(launch_scene_bins(Bn, E));
            RC(dbw_launch_work_scatter(wf, B, H, W, E));
            RC(signal(E, F_SCATTER, p->ev_scatter));
        } else {
            A.scene0 = 0; A.nscenes = 2; Bn.scene0 = 0; Bn.nscenes = 2;
            if (prologue_signal_folded) { A.sync_flag = p->sync_words + F_PROLOGUE; A.sync_val = ++p->sync_val[F_PROLOGUE]; }
            RC(launch_scene_setup(A, M));
            unsigned *tex_flag = nullptr, tex_val = 0;
            if (tex_in_bins) {
                for (int i = 0; i < 3; ++i) Bn.tex.s[i] = sets[i];
                Bn.tex.s[3] = sets[0];
                Bn.tex_sets = 3;
                const long long per = (long long)Bn.nx * Bn.ny * B;
                Bn.tex_z = (int)((2048 + per - 1) / per);
                tex_flag = p->sync_words + F_TEX; tex_val = ++p->sync_val[F_TEX];
            }
            RC(launch_scene_bins(Bn, M));
            RC(dbw_launch_work_scatter(wf, B, H, W, M, tex_flag, tex_val));
        }
    } else {
        RC(dbw_project_clip_fwd(d.env_verts, d.env_faces, in->R, in->T, d.Kmat, B, Ve, Fe, d.cam_eps, zc_on, d.z_clip, d.perspective_correct, FP(L.e.fvc),
                                IP(L.e.first), IP(L.e.num), IP(L.e.c2o), IP(L.e.nbr), IP(L.e.code), FP(L.e.cw), M));
        RC(dbw_project_clip_fwd(FP(L.blk_verts), d.block_faces, in->R, in->T, d.Kmat, B, Vf, Ff, d.cam_eps, zc_on, d.z_clip, d.perspective_correct,
                                FP(L.f.fvc), IP(L.f.first), IP(L.f.num), IP(L.f.c2o), IP(L.f.nbr), IP(L.f.code), FP(L.f.cw), M));
        RC(dbw_render_fwd_fused(FP(L.e.fvc), IP(L.e.first), IP(L.e.num), IP(L.e.nbr), IP(L.e.c2o), IP(L.e.code), FP(L.e.cw), 2 * Fe, d.env_face_uvs,
                                d.env_face_map, d.env_map_desc, FP(L.env_maps), nullptr, 0, B, Fte, H, W, 1, Fe, 0.f, 0.f, d.perspective_correct, d.bg_env,
                                IP(L.p2f_e), FP(L.bary_e), FP(L.dists_e), FP(L.img_e), ws + L.e.rws, L.e.rws_bytes, 3, 1, 1, M));
        RC(dbw_render_fwd_fused_mse(FP(L.f.fvc), IP(L.f.first), IP(L.f.num), IP(L.f.nbr), IP(L.f.c2o), IP(L.f.code), FP(L.f.cw), 2 * Ff, d.block_face_uvs,
                                    d.block_face_map, d.block_map_desc, FP(L.blk_maps), fa, alpha_len, B, Ftf, H, W, K, Ff, d.sigma, d.blur_radius,
                                    d.perspective_correct, d.bg_fg, IP(L.p2f), FP(L.bary), FP(L.dists), ws + L.f.rws, L.f.rws_bytes, nullptr, nullptr, 0.f,
                                    nullptr, nullptr, nullptr, 1, 1, M));
    }

    if (prologue_signal_folded) RC(side_head());

    // ---- M: the env pass (hard, one face per pixel), then the fg pass ending in the composite + MSE ----
#define PROF(i, st) do { if (p->profile) HIP_OK(hipEventRecord(p->ev_t[i], st)); } while (0)
    PROF(0, M);
    if (!fold)
        RC(dbw_render_fwd_fused(FP(L.e.fvc), IP(L.e.first), IP(L.e.num), IP(L.e.nbr), IP(L.e.c2o), IP(L.e.code), FP(L.e.cw), 2 * Fe, d.env_face_uvs,
                                d.env_face_map, d.env_map_desc, FP(L.env_maps), nullptr, 0, B, Fte, H, W, 1, Fe, 0.f, 0.f, d.perspective_correct, d.bg_env,
                                IP(L.p2f_e), FP(L.bary_e), FP(L.dists_e), FP(L.img_e), ws + L.e.rws, L.e.rws_bytes, 3, 2, 1, M));
    PROF(1, M);
    if (setup_aside) RC(await(M, F_SCATTER, p->ev_scatter));
    PROF(2, M);
    if (fold) {
        // the fg pass with the env layer inside it: no env pass, no env image; the env scene's hard uv-fragments leave from here
        EnvFoldHost fh;
        fh.ws = &we; fh.first_idx = IP(L.e.first); fh.num_faces = IP(L.e.num); fh.maps = FP(L.env_maps);
        for (int i = 0; i < 3; ++i) fh.bg[i] = d.bg_env[i];
        fh.p2f = IP(L.p2f_e); fh.uvj = FP(L.bary_e);
        RC(render_fwd_fused_mse_fold(FP(L.f.fvc), IP(L.f.first), IP(L.f.num), IP(L.f.nbr), IP(L.f.c2o), IP(L.f.code), FP(L.f.cw), 2 * Ff, d.block_face_uvs,
                                     d.block_face_map, d.block_map_desc, FP(L.blk_maps), fa, alpha_len, B, Ftf, H, W, K, Ff, d.sigma, d.blur_radius,
                                     d.perspective_correct, d.bg_fg, IP(L.p2f), FP(L.bary), FP(L.dists), ws + L.f.rws, L.f.rws_bytes, target, mse_scale,
                                     FP(L.part), FP(L.g_fg), FP(L.g_env), fh, phase == 1 ? in->rec_out : nullptr, phase == 2 ? in->grad_rec : nullptr, M));
        if (phase == 1) { p->phase1_done = true; return DBW_OK; }      // the caller's term on rec_out, then phase 2
        p->phase1_done = false;
    } else
    RC(dbw_render_fwd_fused_mse(FP(L.f.fvc), IP(L.f.first), IP(L.f.num), IP(L.f.nbr), IP(L.f.c2o), IP(L.f.code), FP(L.f.cw), 2 * Ff, d.block_face_uvs,
                                d.block_face_map, d.block_map_desc, FP(L.blk_maps), fa, alpha_len, B, Ftf, H, W, K, Ff, d.sigma, d.blur_radius,
                                d.perspective_correct, d.bg_fg, IP(L.p2f), FP(L.bary), FP(L.dists), ws + L.f.rws, L.f.rws_bytes, FP(L.img_e), target, mse_scale,
                                FP(L.part), FP(L.g_fg), FP(L.g_env), 2, 1, M));
    PROF(3, M);
    const bool seq = d.backward_order != 0 || (bins && !d.binned_concurrent);     // the env chain waits for the fg backward KERNEL
    // the fg pass's signal to Rg (loss values) and E (env backward): carried by the first workgroup of the fg backward, M's next kernel -- the two
    // streams' work is then enqueued behind that launch (a poll never in front of its producer); with events: recorded here, as ever
    const bool fwd_signal_folded = flags && two;
    if (two && !fwd_signal_folded) RC(signal(M, F_FG_FWD, p->ev_fg_fwd));

    // ---- Rg: the loss values (nothing is differentiated through them) ----
    auto loss_values = [&](hipStream_t st) -> int {
        hipLaunchKernelGGL(loss_finish_kernel, dim3(LOSS_BLOCKS), dim3(256), 0, st, FP(L.part), (long long)B * L.tiles, mse_scale,
                           d.tv_value_scale != 0.f ? d.tv_value_scale : 1.f, vals, FP(L.losses));
        RC(dbw_check_launch("loss_finish_kernel"));
        // everything of Rg that the other streams wait for is done here: the copy to the host (of values outside the zero arena) is nobody's
        // business but the host's -- behind the signal, so that neither the env chain nor Adam ever waits for a transfer
        if (two) RC(signal(st, F_REG, p->ev_reg));
        if (in->read_losses) {
            HIP_OK(hipMemcpyAsync(p->host_losses, FP(L.losses), 5 * sizeof(float), hipMemcpyDeviceToHost, st));
            HIP_OK(hipEventRecord(p->ev_losses, st));
            p->losses_pending = true;
        }
        return DBW_OK;
    };

    // ---- E: backward of the env pass and its tail ----
    auto env_backward = [&](hipStream_t st) -> int {
        PROF(6, st);
        RC(dbw_render_bwd_fused(IP(L.p2f_e), FP(L.bary_e), FP(L.dists_e), IP(L.e.c2o), IP(L.e.code), FP(L.e.cw), 2 * Fe, d.env_face_uvs, d.env_face_map,
                                d.env_map_desc, FP(L.env_maps), nullptr, 0, B, H, W, 1, Fe, 0.f, d.bg_env, FP(L.g_env), FP(L.e.fvc), d.perspective_correct, 0,
                                FP(L.g_env_maps), nullptr, FP(L.g_fvc_e), 1, 3, nullptr, nullptr, nullptr, 0, nullptr, d.n_sky_faces, nullptr, 1, st));
        PROF(7, st);
        RC(dbw_project_clip_bwd(d.env_verts, d.env_faces, in->R, in->T, d.Kmat, B, Ve, Fe, d.cam_eps, d.z_clip, d.perspective_correct, IP(L.e.num),
                                IP(L.e.c2o), IP(L.e.code), FP(L.e.cw), FP(L.g_fvc_e), FP(L.g_env_verts), st));
        RC(dbw_posed_mesh_bwd(d.ground_base, d.n_ground_verts, d.R6_ground, d.T_ground, d.S_world, d.R_world, FP(L.g_env_verts) + (size_t)d.n_sky_verts * 3,
                              d.g_R6_ground, d.g_T_ground, st));
        if (defer) return DBW_OK;
        // (Rg: the TV gradients of the sky / ground maps -- and, for M behind this chain: d / d alpha_full, the pose gradients of the overlap term)
        if (two) RC(await(st, F_REG, p->ev_reg));
        dbw_texture_set env_sets[2] = {sets[0], sets[2]};
        if (!tv) { env_sets[0].grad_sig = nullptr; env_sets[1].grad_sig = nullptr; }
        RC(dbw_texture_prep_bwd_sets(env_sets, 2, st));
        return DBW_OK;
    };
    auto side_after_fwd = [&]() -> int {
        if (two) {
            RC(await(Rg, F_FG_FWD, p->ev_fg_fwd));
            RC(loss_values(Rg));
        }
        if (two && !seq) {
            RC(await(E, F_FG_FWD, p->ev_fg_fwd));
            RC(env_backward(E));
            RC(signal(E, F_ENV_DONE, p->ev_env_done));
        }
        return DBW_OK;
    };
    if (!fwd_signal_folded) RC(side_after_fwd());

    // ---- M: backward of the fg pass and its tail ----
    if (bins && two) RC(await(M, F_LAYOUT, p->ev_layout));      // (this step's cursors and sub-ranges come from Rg)
    PROF(4, M);
    unsigned *bwd_flag = nullptr, bwd_val = 0;
    if (fwd_signal_folded) { bwd_flag = p->sync_words + F_FG_FWD; bwd_val = ++p->sync_val[F_FG_FWD]; }
    RC(render_bwd_fused_signal(IP(L.p2f), FP(L.bary), FP(L.dists), IP(L.f.c2o), IP(L.f.code), FP(L.f.cw), 2 * Ff, d.block_face_uvs, d.block_face_map,
                               d.block_map_desc, FP(L.blk_maps), fa, alpha_len, B, H, W, K, Ff, d.sigma, d.bg_fg, FP(L.g_fg), FP(L.f.fvc), d.perspective_correct, 1,
                               FP(L.g_blk_maps), coarse ? FP(L.g_fa) : nullptr, FP(L.g_fvc_f), d.decim_blocks > 1 ? 1 : 0, 2, bins ? d.block_bin_base : nullptr, cursor,
                               bins ? (void *)(ws + L.records) : nullptr, bins ? L.bin_cap : 0, blayout, 0, nullptr, 1, M, bwd_flag, bwd_val));
    PROF(5, M);
    if (fwd_signal_folded) RC(side_after_fwd());
    if (two && seq) {
        RC(signal(M, F_KERNEL_DONE, p->ev_kernel_done));
        RC(await(E, F_KERNEL_DONE, p->ev_kernel_done));
        RC(env_backward(E));
        RC(signal(E, F_ENV_DONE, p->ev_env_done));
    }
    if (bins) RC(dbw_texbin_reduce(d.block_bin_info, cursor, ws + L.records, L.bin_cap, blayout, d.n_bins, FP(L.g_blk_maps), M));
    // the backward of the blocks' texture preparation, first in the tail: a data-parallel caller reduces the blocks' texture gradient -- 83 %
    // of the gradient bytes -- as soon as ev_blocks_ready says so, next to everything below.  It needs the TV gradient of the blocks' maps
    // (Rg): data parallel M waits for it here; on one GPU the launch moves behind the join with the env chain (which has waited for Rg), so
    // that M pays for one wait instead of two
    const bool early_textures = two && !in->with_adam && !defer;
    auto blocks_textures = [&]() -> int {
        if (defer) return DBW_OK;
        dbw_texture_set blk = sets[1];
        if (!tv) blk.grad_sig = nullptr;
        RC(dbw_texture_prep_bwd_sets(&blk, 1, M));
        if (early_textures) RC(signal(M, F_BLOCKS_READY, p->ev_blocks_ready));      // (an event costs M ~7 us: only where somebody waits for it)
        return DBW_OK;
    };
    if (early_textures) {
        RC(await(M, F_REG, p->ev_reg));
        RC(blocks_textures());
    }
    // fuse bit 5: the blocks' projection backward and the backward of their texture preparation -- independent of each other -- in ONE launch
    // behind the join, instead of one in front of it and one behind (a dependent launch less on the critical chain)
    const bool tail_merged = (d.fuse & 32) && !early_textures && !defer;
    if (!tail_merged)
        RC(dbw_project_clip_bwd(FP(L.blk_verts), d.block_faces, in->R, in->T, d.Kmat, B, Vf, Ff, d.cam_eps, d.z_clip, d.perspective_correct, IP(L.f.num),
                                IP(L.f.c2o), IP(L.f.code), FP(L.f.cw), FP(L.g_fvc_f), FP(L.g_blk_verts), M));
    if (!two) { RC(env_backward(M)); RC(loss_values(M)); }
    if (two) RC(await(M, F_ENV_DONE, p->ev_env_done));           // the env chain, and through it the regularisers (E waited for Rg)
    if (two && defer) RC(await(M, F_REG, p->ev_reg));            // (... which it did not when the texture tail is deferred)
    if (tail_merged) {
        ClipBwdArgs C;
        C.verts = FP(L.blk_verts); C.faces = d.block_faces; C.R = in->R; C.T = in->T; C.Kmat = d.Kmat; C.B = B; C.V = Vf; C.F = Ff;
        C.eps = d.cam_eps; C.zc = d.z_clip; C.persp = d.perspective_correct;
        C.num_faces = IP(L.f.num); C.c2o = IP(L.f.c2o); C.code = IP(L.f.code); C.cw = FP(L.f.cw); C.gfvc = FP(L.g_fvc_f); C.gverts = FP(L.g_blk_verts);
        dbw_texture_set blk = sets[1];
        if (!tv) blk.grad_sig = nullptr;
        RC(launch_clip_bwd_tex(C, blk, M));
    } else if (!early_textures) RC(blocks_textures());
    if ((d.fuse & 8) && (d.fuse & 1)) {
        BlocksTailArgs A;
        memset(&A, 0, sizeof(A));
        A.sq_eps = d.sq_eps; A.S = d.S; A.R6 = d.R6; A.T = d.T; A.sq_local = FP(L.sq_local); A.keep = IP(L.keep); A.nb = nb; A.nv = nv;
        A.scale_min = d.scale_min; A.S_world = d.S_world; A.Rw = d.R_world; A.g_verts = FP(L.g_blk_verts);
        A.g_sq_eps = d.g_sq_eps; A.g_S = d.g_S; A.g_R6 = d.g_R6; A.g_T = d.g_T;
        A.alpha = FP(L.alpha); A.g_alpha_parts = coarse ? FP(L.g_fa) : nullptr; A.alpha_parts = 64; A.g_alpha_full = FP(L.g_alpha_full); A.g_logit = d.g_alpha_logit;
        A.void_raised = void_raised; A.void_flag = void_flag;          // (the run's latch: behind the join, in front of Adam)
        RC(launch_blocks_tail(A, M));
    } else {
        RC(dbw_sq_blocks_bwd(d.sq_eps, d.S, d.R6, d.T, d.trig, IP(L.keep), 0, nb, nv, d.ratio_block_scene, d.scale_min, d.S_world, d.R_world, FP(L.g_blk_verts),
                             d.g_sq_eps, d.g_S, d.g_R6, d.g_T, M));
        RC(dbw_block_alpha_bwd(FP(L.alpha), IP(L.keep), coarse ? FP(L.g_fa) : nullptr, 64, FP(L.g_alpha_full), nb, d.g_alpha_logit, M));
        hipLaunchKernelGGL(void_latch_kernel, dim3(1), dim3(1), 0, M, (const float *)void_raised, void_flag);
        RC(dbw_check_launch("void_latch_kernel"));
    }

    // ---- M: Adam on both learning-rate groups, which also clears the zero arena for the next run ----
    if (in->with_adam) {
        RC(dbw_adam_step_groups(d.flat_param, d.flat_grad, d.exp_avg, d.exp_avg_sq, d.group_end, in->lr, 2, in->beta1, in->beta2, in->adam_eps, in->adam_step,
                                ws + L.arena_begin, (int64_t)(L.arena_end - L.arena_begin), void_flag, M));
        p->arena_clean = true;
    }
    if (bins) { p->bin_turn = 1 - p->bin_turn; p->bin_ready = 1; }
    p->runs += 1;
    p->profiled = p->profile;
    return DBW_OK;
#undef PROF
#undef FP
#undef IP
}

extern "C" int dbw_train_step_finish(dbw_step_plan *p, const dbw_step_inputs *in, dbw_stream_t stream) {
    DBW_REQUIRE(p && in, "null pointer");
    DBW_REQUIRE(!in->with_adam || in->adam_step >= 1, "adam_step >= 1");
    const dbw_step_desc &d = p->d;
    const Layout &L = p->L;
    hipStream_t M = (hipStream_t)stream;
    dbw_texture_set sets[3];
    fill_texture_sets(d, L, p->ws, sets);
    if (tv_on(d)) add_tv_fields(L, p->ws, sets);
    RC(dbw_texture_prep_bwd_sets(sets, 3, M));
    if (in->with_adam) {
        RC(dbw_adam_step_groups(d.flat_param, d.flat_grad, d.exp_avg, d.exp_avg_sq, d.group_end, in->lr, 2, in->beta1, in->beta2, in->adam_eps, in->adam_step,
                                p->ws + L.arena_begin, (int64_t)(L.arena_end - L.arena_begin), (const float *)(p->ws + L.losses) + 7, M));
        p->arena_clean = true;
    }
    return DBW_OK;
}

extern "C" int dbw_train_step_wait_blocks_ready(dbw_step_plan *p, dbw_stream_t stream) {
    DBW_REQUIRE(p, "null pointer");
    if (!p->cur_flags) { HIP_OK(hipStreamWaitEvent((hipStream_t)stream, p->ev_blocks_ready, 0)); return DBW_OK; }
    hipLaunchKernelGGL(sync_wait_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const unsigned *)(p->sync_words + F_BLOCKS_READY), p->sync_val[F_BLOCKS_READY],
                       p->sync_words + SYNC_TIMEOUT_SLOT, p->host_timeouts_dev, (float *)(p->sync_words + SYNC_VOID_SLOT), SYNC_LIMIT_TICKS);
    return dbw_check_launch("sync_wait_kernel");
}

// (tests: the join of the NEXT run polls for a value that never comes and gives up after 0.05 s -- the real thing, end to end)
extern "C" int dbw_debug_train_step_force_timeout(dbw_step_plan *p) {
    if (!p) return DBW_ERR_INVALID;
    p->force_timeout = 1;
    return DBW_OK;
}
// (tests: the side streams' polls for the prologue of the NEXT run give up after 0.02 s -- with their real value: behind a main stream that
// the caller has stalled for longer than that, the side streams then do run ahead of the prologue)
extern "C" int dbw_debug_train_step_hasty_prologue_wait(dbw_step_plan *p) {
    if (!p) return DBW_ERR_INVALID;
    p->force_timeout = 2;
    return DBW_OK;
}

// tests / diagnostics: out3 = {index of the counter the first poll that gave up waited on (enum F_* above; -1: none gave up), the value it
// wanted, the value it last saw}; synchronises the device
extern "C" int dbw_debug_train_step_last_timeout(dbw_step_plan *p, int *out3) {
    DBW_REQUIRE(p && out3, "null pointer");
    unsigned w[3] = {0u, 0u, 0u};
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(w, p->sync_words + SYNC_TIMEOUT_SLOT - 3, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) {
        dbw_set_error("dbw_debug_train_step_last_timeout: copy failed");
        return DBW_ERR_LAUNCH;
    }
    out3[0] = (int)w[0] - 1; out3[1] = (int)w[1]; out3[2] = (int)w[2];
    return DBW_OK;
}
// ... and all twelve counters: as that poll saw them when it gave up (now, if none did), and the values the host has asked for so far
extern "C" int dbw_debug_train_step_counters(dbw_step_plan *p, unsigned *seen12, unsigned *asked12) {
    DBW_REQUIRE(p && seen12 && asked12, "null pointer");
    unsigned first = 0u;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&first, p->sync_words + SYNC_TIMEOUT_SLOT - 3, sizeof(first), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(seen12, p->sync_words + (first ? SYNC_TIMEOUT_SLOT + 1 : 0), SYNC_FLAGS * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) {
        dbw_set_error("dbw_debug_train_step_counters: copy failed");
        return DBW_ERR_LAUNCH;
    }
    for (int i = 0; i < SYNC_FLAGS; ++i) asked12[i] = p->sync_val[i];
    return DBW_OK;
}
extern "C" int dbw_train_step_voided_runs(const dbw_step_plan *p) { return p ? p->voided_runs + (*(volatile unsigned *)p->host_timeouts != 0u ? 1 : 0) : -1; }

extern "C" int64_t dbw_train_step_void_flag_offset(const dbw_step_plan *p) { return p ? (int64_t)p->L.losses + 7 * (int64_t)sizeof(float) : -1; }

extern "C" int dbw_train_step_sync_timeouts(dbw_step_plan *p) {
    if (!p) return -1;
    unsigned n = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&n, p->sync_words + SYNC_TIMEOUT_SLOT, sizeof(n), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)n;
}

extern "C" int dbw_train_step_profile(dbw_step_plan *p, int on) {
    DBW_REQUIRE(p, "null pointer");
    p->profile = on != 0;
    p->profiled = false;
    return DBW_OK;
}

extern "C" int dbw_train_step_kernel_times(dbw_step_plan *p, float *out4_ms) {
    DBW_REQUIRE(p && out4_ms, "null pointer");
    DBW_REQUIRE(p->profiled, "no profiled run: dbw_train_step_profile(plan, 1), then a run");
    for (int k = 0; k < 4; ++k) {
        HIP_OK(hipEventSynchronize(p->ev_t[2 * k + 1]));
        HIP_OK(hipEventElapsedTime(out4_ms + k, p->ev_t[2 * k], p->ev_t[2 * k + 1]));
    }
    return DBW_OK;
}
