// C-ABI of libltgpu.so (include/lt_gpu.h): context, framebuffer and the entry points of
// the polygon path.  Host code only; the kernels are in lt_kernels.hip.
#include "../../include/lt_gpu.h"
#include "lt_kernels.h"

#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace lt;

namespace {

thread_local std::string g_create_error;

// batches of fewer polygons fill in draw order by reading every header in every band
constexpr int BAND_LIST_MIN_POLYS = 20000;

// small device scratch: order flag and the band lists' allocation word (first 256 bytes), then the setup sub-queue counters
constexpr size_t kScratchBytes = 256 + 2 * (size_t)QUEUES * QUEUE_COUNT_STRIDE * sizeof(unsigned);

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

// HIP reads GPU_MAX_HW_QUEUES (hardware queues per process, four if unset) when its runtime starts,
// which is at the process's first HIP call, not when its library is loaded.  A program that loads
// this library before that call and has not set the variable gets eight, the number the library's
// four streams beside the caller's need to run side by side; a value already in the environment
// stays.  Once the runtime has started the variable is past changing: the runtime holds the
// kernel driver's device node open from then on, and a process that already has it is left alone.
__attribute__((constructor)) void default_hw_queues()
{
    if (getenv("GPU_MAX_HW_QUEUES")) return;
    for (int fd = 0; fd < 1024; fd++) {
        char link[32], target[64];
        snprintf(link, sizeof link, "/proc/self/fd/%d", fd);
        const ssize_t n = readlink(link, target, sizeof target - 1);
        if (n <= 0) continue;
        target[n] = 0;
        if (!strcmp(target, "/dev/kfd")) return;
    }
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
}

}  // namespace

// Work arrays of lt_submit_object_dev_noalloc: one array per GPU, held from the moment the
// library's code is loaded on it until the process ends.  In host code the name below is only
// a handle for hipGetSymbolAddress, never a device address: lt_ctx_create asks once per device
// for the array's address there and every later use goes through that pointer.
static __device__ __align__(256) unsigned char scratch[LT_NOALLOC_SCRATCH_BYTES];

namespace {

struct NoAllocDevice {
    void *ptr = nullptr;               // address of `scratch` on this device
    hipEvent_t done = nullptr;         // end of the last batch that used it
    bool pending = false;
};
NoAllocDevice g_noalloc[MAX_DEVICES];
std::mutex g_noalloc_mutex;

// once per device, with that device current
hipError_t noalloc_init(int device)
{
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    NoAllocDevice &d = g_noalloc[device];
    if (d.ptr) return hipSuccess;
    void *p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    if (!d.done && (e = hipEventCreateWithFlags(&d.done, hipEventDisableTiming)) != hipSuccess) return e;
    d.ptr = p;
    return hipSuccess;
}

}  // namespace

// Host arrays on their way to the device.  HIP copies pageable memory through a bounce buffer of
// its own on one thread (30 GB/s measured for 160 MB); the same arrays copied by four threads into
// a ring of two pinned 16 MB pieces, each piece sent while the next is being filled, arrive at
// 50 GB/s (pinned memory itself: 57).  The ring, its events and the three helper threads belong to
// the context and exist from its first large host-array call until lt_ctx_destroy.
struct HostStage {
    static constexpr size_t PIECE = (size_t)16 << 20;
    static constexpr int MAX_HELPERS = 15;
    int HELPERS = 3;                               // lt_set_copy_threads - 1
    char *ring = nullptr;
    hipEvent_t sent[2] = {nullptr, nullptr};      // the copy out of a piece has finished
    bool in_flight[2] = {false, false};
    unsigned next = 0;
    std::vector<std::thread> helpers;
    std::mutex mu;
    std::condition_variable go, done;
    char *dst = nullptr;
    const char *src = nullptr;
    size_t bytes = 0;
    unsigned generation = 0;
    int left = 0;
    bool quit = false;

    void part_of(size_t n, int part, size_t &a, size_t &b)
    {
        a = (n * (size_t)part / (HELPERS + 1)) & ~(size_t)63;
        b = part == HELPERS ? n : (n * (size_t)(part + 1) / (HELPERS + 1)) & ~(size_t)63;
    }
    // `seen`: the generation at the time the thread was made (its maker is the only one to advance it)
    void helper(int part, unsigned seen)
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            go.wait(lk, [&] { return quit || generation != seen; });
            if (quit) return;
            seen = generation;
            char *d = dst;
            const char *s = src;
            size_t a, b;
            part_of(bytes, part, a, b);
            lk.unlock();
            memcpy(d + a, s + a, b - a);
            lk.lock();
            if (--left == 0) done.notify_one();
        }
    }
    // one piece copied by the helpers and the calling thread together
    void spread(char *d, const char *s, size_t n)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            dst = d; src = s; bytes = n; left = HELPERS; generation++;
        }
        go.notify_all();
        size_t a, b;
        part_of(n, 0, a, b);
        memcpy(d + a, s + a, b - a);
        std::unique_lock<std::mutex> lk(mu);
        done.wait(lk, [&] { return left == 0; });
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        go.notify_all();
        for (std::thread &t : helpers) t.join();
        helpers.clear();
        quit = false;
    }
};

struct lt_ctx {
    int device = 0;
    int W = 0, H = 0, tiles_x = 0, tiles_y = 0, ntiles = 0;     // tile counts for the narrow tile (the larger number)
    int tile_mode = 0;                             // 0: chosen from what the last batches drew; else TW_NARROW / TW_WIDE
    bool auto_wide = false;                        // the choice when tile_mode == 0
    bool fragment_heavy = false;                   // the last look found >= 256 depth tests a polygon: batches are not overlapped
    unsigned long long seen_fragments = 0, seen_drawn = 0;
    int tile_waves = 0;                            // 0: chosen from the last batches' longest tile lists; else 4 / 8 waves on a 64-pixel tile
    bool auto_waves8 = false;                      // the choice when tile_waves == 0
    unsigned long long seen_longest = 0, seen_refs = 0;
    unsigned seen_batches = 0;
    hipStream_t stream = nullptr;
    int *cbuf = nullptr, *zbuf = nullptr;          // the frame being drawn (own or bound)
    int *own_cbuf = nullptr, *own_zbuf = nullptr;  // allocated by lt_ctx_create
    // Two frames in flight: a cleared frame depends on nothing drawn before it, so with the
    // context's own buffers every lt_clear that follows a draw moves to the other of two
    // buffer pairs and the other of two fill streams, and the new frame's fills need not wait
    // for the old frame's (whose last workgroups leave part of the chip idle).  Off while a
    // caller's frame is bound, while a band is set, once lt_framebuffer has handed out the
    // buffers' addresses, and with lt_set_frame_overlap(ctx, 0).
    int *alt_cbuf = nullptr, *alt_zbuf = nullptr;  // second pair, allocated at its first use
    hipStream_t alt_rs = nullptr;                  // the second pair's fill stream
    hipEvent_t ev_side = nullptr;
    bool frame_overlap = true, fb_exposed = false;
    bool on_alt = false;                           // which of the two own pairs holds the current own frame
    bool drawn_since_clear = false;
    int band_y0 = 0, band_y1 = 0;                  // rows this context draws
    ClipWindow clip{0, 0, 0, 0};                   // C_x_clipping_min .. C_y_clipping_max for the batches submitted next
    bool clear_pending = false;
    int clear_colour = 0;
    bool frame_defined = false;
    Status *status = nullptr;          // device
    Status base{};                     // host copy at the last reset
    uint64_t polys_in = 0;
    uint64_t list_capacity_user = 0;
    int order_mode = 0;                // LT_ORDER_AUTO / ALWAYS / NEVER
    int order_lists = 0;               // 0: band lists for large batches; LT_ORDER_LISTS_ALWAYS / LT_ORDER_LISTS_NEVER
    // Per-batch workspace (grown on demand, reused).  The set-up and binning of a batch (which
    // do not touch the frame) run on a geometry stream while the fill of the batch before it
    // runs on `rs`.  Small batches take the sets in turns of four and the two geometry streams
    // in turns of two, so that the set-up chains of two batches are in flight at once (the
    // short binning kernels of one leave most of the chip to the other); batches whose work
    // arrays are large stay with two sets and one geometry stream.
    struct Work {
        DevBuf hdr, edges, prp, list, tcount, toffset, tcursor, queue, scratch;   // scratch: order flag + sub-queue counters
        DevBuf bcnt, bplace, blist;                                               // band lists of the draw-order fill
        DevBuf sq_desc, sq_tuples, sq_queue;                                      // screen queue of the fast set-up kernel
        hipEvent_t geom_done = nullptr, raster_done = nullptr;
        bool in_use = false;
    } work[4];
    int last_set = 0;                  // the set the batch before took
    bool fill_shares_chip = false;     // the caller submits faster than the chip draws: this batch's fill will have other batches' set-up beside it
    unsigned batch_no = 0;
    bool pipelined = true;
    hipStream_t own_gs = nullptr, own_gs2 = nullptr, own_rs = nullptr;    // internal streams (non-blocking)
    hipStream_t gs = nullptr, rs = nullptr;            // streams in use: own_* or `stream`
    hipEvent_t ev_fork = nullptr;
    DevBuf s_verts, s_polys, s_aux, s_pv, s_normals, s_pn, s_prp, s_objects, s_polyobj;      // staging for host-pointer calls
    DevBuf line_winner, line_colours, s_lines;   // wireframe: per-pixel winning line, per-polygon colour, staged lines
    HostStage host_stage;
    // textures
    int *tex_data[LT_MAX_TEXTURES] = {};
    int tex_log_h[LT_MAX_TEXTURES] = {};
    const int **tex_ptr_d = nullptr;
    int *tex_log_d = nullptr;
    // stage timing
    int profiling = 0;                 // bit k: stage k is timed
    unsigned profile_every = 1;        // ... in every profile_every-th batch
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    struct Span { int stage; size_t begin, end; };      // indices into ev_pool
    std::vector<Span> spans;
    int timed_batches = 0;
    double stage_ms[LT_STAGES] = {};
    int stage_submits = 0;
    std::string err;
};

namespace {

int fail(lt_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, call)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) return fail(c, LT_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

int sync_all(lt_ctx *c);

int ensure(lt_ctx *c, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return LT_OK;
    if (b.p) {
        int rc = sync_all(c);           // nothing queued may still use the old array
        if (rc) return rc;
        HIPCHK(c, hipFree(b.p));
        b.p = nullptr; b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    HIPCHK(c, hipMalloc(&b.p, want));
    b.cap = want;
    return LT_OK;
}

// A host array to device memory on the caller's stream.  On return the source has been read,
// unless it is pinned memory of the caller's (hipHostMalloc / hipHostRegister), which the copy
// engine reads by itself: every caller of this waits for the stream before it returns.
int h2d(lt_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return LT_OK;
    bool plain = bytes < ((size_t)1 << 20);
    if (!plain) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, src) == hipSuccess) plain = at.type == hipMemoryTypeHost;
        else (void)hipGetLastError();          // an address HIP has never seen: not an error here
    }
    if (plain) {
        HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        return LT_OK;
    }
    HostStage &h = c->host_stage;
    if (!h.ring) {
        HIPCHK(c, hipHostMalloc((void **)&h.ring, 2 * HostStage::PIECE, hipHostMallocDefault));
        for (hipEvent_t &e : h.sent) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (h.helpers.empty())
        for (int i = 1; i <= h.HELPERS; i++) h.helpers.emplace_back([&h, i, g = h.generation] { h.helper(i, g); });
    for (size_t off = 0; off < bytes; off += HostStage::PIECE) {
        const size_t n = bytes - off < HostStage::PIECE ? bytes - off : HostStage::PIECE;
        const unsigned k = h.next++ & 1u;
        char *piece = h.ring + k * HostStage::PIECE;
        if (h.in_flight[k]) HIPCHK(c, hipEventSynchronize(h.sent[k]));
        h.spread(piece, (const char *)src + off, n);
        HIPCHK(c, hipMemcpyAsync((char *)dst + off, piece, n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipEventRecord(h.sent[k], c->stream));
        h.in_flight[k] = true;
    }
    return LT_OK;
}

// Device memory to a host array, through the same ring in the other direction: a piece is copied
// out by the threads while the copy engine fills the other one.  The host array holds the data on
// return.  Work queued on the caller's stream before this call is waited for, as a copy on that
// stream would.
int d2h(lt_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return LT_OK;
    bool plain = bytes < ((size_t)1 << 20);
    if (!plain) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, dst) == hipSuccess) plain = at.type == hipMemoryTypeHost;
        else (void)hipGetLastError();
    }
    HostStage &h = c->host_stage;
    if (!plain && !h.ring) {
        HIPCHK(c, hipHostMalloc((void **)&h.ring, 2 * HostStage::PIECE, hipHostMallocDefault));
        for (hipEvent_t &e : h.sent) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (plain) {
        HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return LT_OK;
    }
    if (h.helpers.empty())
        for (int i = 1; i <= h.HELPERS; i++) h.helpers.emplace_back([&h, i, g = h.generation] { h.helper(i, g); });
    // pieces of the ring that a host-to-device copy still reads
    for (int k = 0; k < 2; k++)
        if (h.in_flight[k]) { HIPCHK(c, hipEventSynchronize(h.sent[k])); h.in_flight[k] = false; }
    size_t prev_off = 0, prev_n = 0;
    unsigned prev_k = 0;
    for (size_t off = 0; off < bytes; off += HostStage::PIECE) {
        const size_t n = bytes - off < HostStage::PIECE ? bytes - off : HostStage::PIECE;
        const unsigned k = h.next++ & 1u;
        HIPCHK(c, hipMemcpyAsync(h.ring + k * HostStage::PIECE, (const char *)src + off, n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipEventRecord(h.sent[k], c->stream));
        if (prev_n) {
            HIPCHK(c, hipEventSynchronize(h.sent[prev_k]));
            h.spread((char *)dst + prev_off, h.ring + prev_k * HostStage::PIECE, prev_n);
        }
        prev_off = off; prev_n = n; prev_k = k;
    }
    HIPCHK(c, hipEventSynchronize(h.sent[prev_k]));
    h.spread((char *)dst + prev_off, h.ring + prev_k * HostStage::PIECE, prev_n);
    return LT_OK;
}

// Stage timing: a timed stage is bracketed by two events on the stream it runs on;
// events are pooled and reused.  Only the stages named in the profiling mask pay for
// events (an event costs a few microseconds of stream time).
int new_event(lt_ctx *c, hipStream_t s, size_t *idx)
{
    if (c->ev_used == c->ev_pool.size()) {
        hipEvent_t e;
        HIPCHK(c, hipEventCreate(&e));
        c->ev_pool.push_back(e);
    }
    *idx = c->ev_used++;
    if (s != (hipStream_t)-1) HIPCHK(c, hipEventRecord(c->ev_pool[*idx], s));
    return LT_OK;
}

bool timed(const lt_ctx *c, int stage)
{
    return ((c->profiling >> stage) & 1) && c->batch_no % c->profile_every == 0;
}

int stage_begin(lt_ctx *c, int stage, hipStream_t s)
{
    if (!timed(c, stage)) return LT_OK;
    lt_ctx::Span sp{stage, 0, 0};
    // a stage that follows a timed stage on the same stream starts at that stage's end
    if (stage > 0 && stage < 4 && !c->spans.empty() && c->spans.back().stage == stage - 1) sp.begin = c->spans.back().end;
    else { int rc = new_event(c, s, &sp.begin); if (rc) return rc; }
    c->spans.push_back(sp);
    return LT_OK;
}

// `recorded`: the end event has been recorded by the callee (index in *recorded)
int stage_end(lt_ctx *c, int stage, hipStream_t s, const size_t *recorded = nullptr)
{
    if (!timed(c, stage)) return LT_OK;
    if (recorded) { c->spans.back().end = *recorded; return LT_OK; }
    return new_event(c, s, &c->spans.back().end);
}

// fold the recorded spans into stage_ms; every stream must be idle
int collect(lt_ctx *c)
{
    for (const lt_ctx::Span &sp : c->spans) {
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev_pool[sp.begin], c->ev_pool[sp.end]));
        c->stage_ms[sp.stage] += ms;
    }
    c->stage_submits += c->timed_batches;
    c->timed_batches = 0;
    c->spans.clear();
    c->ev_used = 0;
    return LT_OK;
}

// host waits until everything queued by this context has run
int sync_all(lt_ctx *c)
{
    HIPCHK(c, hipStreamSynchronize(c->gs));
    if (c->own_gs2) HIPCHK(c, hipStreamSynchronize(c->own_gs2));
    HIPCHK(c, hipStreamSynchronize(c->rs));
    if (c->alt_rs) HIPCHK(c, hipStreamSynchronize(c->alt_rs));
    if (c->own_rs) HIPCHK(c, hipStreamSynchronize(c->own_rs));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LT_OK;
}

// sum of the device status slots; the stream must be idle
int read_status(lt_ctx *c, Status *out)
{
    static_assert(sizeof(Status) == 9 * sizeof(unsigned long long), "Status is nine counters");
    std::vector<Status> slots(STATUS_SLOTS);
    HIPCHK(c, hipMemcpy(slots.data(), c->status, sizeof(Status) * STATUS_SLOTS, hipMemcpyDeviceToHost));
    unsigned long long *o = (unsigned long long *)out;
    for (int k = 0; k < 9; k++) o[k] = 0;
    for (const Status &s : slots) {
        const unsigned long long *v = (const unsigned long long *)&s;
        for (int k = 0; k < 9; k++) o[k] += v[k];
    }
    return LT_OK;
}

// as ensure(), and a newly allocated array starts at zero
int ensure_zeroed(lt_ctx *c, DevBuf &b, size_t bytes, hipStream_t first_user)
{
    if (bytes <= b.cap) return LT_OK;
    int rc = ensure(c, b, bytes);
    if (rc) return rc;
    // on the stream whose kernels use the array first: the context's own streams do not
    // wait for the default stream
    HIPCHK(c, hipMemsetAsync(b.p, 0, b.cap, first_user));
    return LT_OK;
}

int use_device(lt_ctx *c)
{
    HIPCHK(c, hipSetDevice(c->device));
    return LT_OK;
}

// Sizes of one batch's work arrays.
struct BatchPlan {
    int emax = 0, max_edges = 0;
    uint64_t capacity = 0;             // tile list entries
    unsigned qcap = 0;                 // entries per set-up sub-queue
    bool want_prp = false;
    size_t hdr = 0, edges = 0, list = 0, prp = 0, tcount = 0, toffset = 0, tcursor = 0, flags = 0;
    size_t queue = 0, sq_queue = 0, sq_desc = 0, sq_tuples = 0;      // object batches only
    // per-band lists of the draw-order fill (textured batches, screens of up to MAX_BANDS bands)
    int band_rows = 0, nbands = 0;
    uint64_t band_capacity = 0;
    size_t bcnt = 0, bplace = 0, blist = 0;
};

// The work arrays themselves: a context's work set, or pieces of the static scratch array.
struct WorkArrays {
    PrimHdr *hdr = nullptr;
    int *edges = nullptr, *prp = nullptr;
    unsigned *list = nullptr, *tcount = nullptr, *toffset = nullptr, *tcursor = nullptr;
    int *flags = nullptr;              // order flag (first 256 bytes), then the set-up sub-queue counters
    uint2 *queue = nullptr, *sq_queue = nullptr;
    int *sq_desc = nullptr, *sq_tuples = nullptr;
    unsigned *bcnt = nullptr, *bplace = nullptr, *blist = nullptr;   // band lists (null: the draw-order fill reads every header)
};

BatchPlan plan_batch(const lt_ctx *c, bool screen, int dim, int np, int max_edges, uint64_t min_capacity, bool screen_prp = false)
{
    BatchPlan p;
    const int na = dim - 2;
    if (max_edges <= 0 || max_edges > 15) max_edges = 15;
    p.max_edges = max_edges;
    p.emax = (screen ? 2 : 4) * max_edges;
    if (p.emax > 15) p.emax = 15;
    p.capacity = c->list_capacity_user ? c->list_capacity_user : (uint64_t)np * 64u;
    if (p.capacity < min_capacity) p.capacity = min_capacity;
    p.qcap = setup_queue_cap(np);
    p.want_prp = (!screen && dim == 8) || screen_prp;
    p.hdr = (size_t)np * sizeof(PrimHdr);
    // one record (header copy + edges 0..2) per slot, then the planes of edges 3 and up
    p.edges = (size_t)np * (16 + (p.emax < 3 ? 3 : p.emax) * edge_ints(na) * sizeof(int));
    p.list = (size_t)p.capacity * sizeof(unsigned);
    p.prp = p.want_prp ? (size_t)np * PRP_INTS * sizeof(int) : 0;
    p.tcount = (size_t)c->ntiles * sizeof(unsigned);
    p.toffset = (size_t)(c->ntiles + 1) * sizeof(unsigned);
    p.tcursor = (size_t)c->ntiles * sizeof(unsigned);
    p.flags = kScratchBytes;
    int R, nb;
    const bool want_lists = c->order_lists == LT_ORDER_LISTS_ALWAYS || (c->order_lists == 0 && np >= BAND_LIST_MIN_POLYS);
    if (dim == 8 && c->order_mode != LT_ORDER_NEVER && want_lists && band_lists_possible(c->W, c->H, &R, &nb)) {
        p.band_rows = R; p.nbands = nb;
        // a polygon of h rows is in h / R + 1 bands at most; eight a polygon, and room for tall ones
        p.band_capacity = c->list_capacity_user ? c->list_capacity_user : (uint64_t)np * 8u + (4u << 20);
        if (p.band_capacity > 0xffffffffull) p.band_capacity = 0xffffffffull;
        p.bcnt = (size_t)((np + 1023) / 1024) * nb * sizeof(unsigned);
        p.bplace = (size_t)2 * nb * sizeof(unsigned);
        p.blist = (size_t)p.band_capacity * sizeof(unsigned);
    }
    if (!screen) {
        p.queue = (size_t)p.qcap * QUEUES * sizeof(uint2);
        p.sq_queue = p.queue;
        p.sq_desc = (size_t)p.qcap * QUEUES * sizeof(int4);
        p.sq_tuples = (size_t)p.qcap * QUEUES * 4 * dim * sizeof(int);
    }
    return p;
}

// set-up -> scan -> bin -> fill for one batch whose polygon data and work arrays are on the
// device.  Set-up and binning are issued on gs, the fill on rs.  geom_done / raster_done (both
// or neither): the hand-over events between the two streams and the "work arrays free" signal.
// The tile counts and the sub-queue counters in `wa` must be zero.
int issue_batch(lt_ctx *c, const WorkArrays &wa, const BatchPlan &pl, hipStream_t gs, hipStream_t rs, hipEvent_t geom_done,
                hipEvent_t raster_done, bool timing, bool screen, int dim, const int32_t *m, const int *d_verts, int nv,
                const int *d_polys, const int *d_aux, int np, const int *d_pv, int npv, int render_type, int log_focus,
                int force_linear, const int *d_prp_src = nullptr, const ObjectDesc *d_objects = nullptr, const int *d_polyobj = nullptr,
                int nobj = 0)
{
    const int na = dim - 2;
    const int emax = pl.emax;
    const uint64_t capacity = pl.capacity;
    const bool want_prp = pl.want_prp;
    const int saved_profiling = c->profiling;
    if (!timing) c->profiling = 0;
    struct Restore { lt_ctx *c; int v; ~Restore() { c->profiling = v; } } restore{c, saved_profiling};
    int rc;
    int *order_flag = wa.flags;

    // the transform is applied inside the setup kernel: its stage is empty
    if ((rc = stage_begin(c, 0, gs))) return rc;
    if ((rc = stage_end(c, 0, gs))) return rc;
    if ((rc = stage_begin(c, 1, gs))) return rc;
    const int order_gen = (int)(c->batch_no % 0x7fffffffu) + 1;

    SetupArgs sa{};
    sa.use_matrix = m ? 1 : 0;
    if (m) for (int i = 0; i < 12; i++) sa.m[i] = m[i];
    sa.verts = d_verts; sa.polys = d_polys; sa.polyaux = d_aux; sa.polyverts = d_pv;
    sa.np = np; sa.nv = nv; sa.npv = npv;
    sa.objects = d_objects; sa.polyobj = d_polyobj; sa.nobj = nobj;
    if (d_objects) sa.use_matrix = 1;
    sa.render_type = render_type; sa.log_focus = log_focus; sa.force_linear = force_linear;
    sa.max_edges = pl.max_edges;
    // tile width of this batch
    const int tile_w = c->tile_mode ? c->tile_mode : (c->auto_wide ? TW_WIDE : TW_NARROW);
    const int twl = tile_w == TW_WIDE ? 7 : 6;
    const int tiles_x = (c->W + tile_w - 1) / tile_w, ntiles = tiles_x * c->tiles_y;
    sa.W = c->W; sa.H = c->H; sa.tiles_x = tiles_x; sa.tiles_y = c->tiles_y;
    sa.tile_w_log = twl;
    sa.emax = emax;
    sa.hdr = wa.hdr; sa.edges = wa.edges;
    sa.prp = want_prp && !screen ? wa.prp : nullptr;
    // screen-level perspective mapper: the caller's constants become this batch's (slot = polygon)
    if (d_prp_src) HIPCHK(c, hipMemcpyAsync(wa.prp, d_prp_src, (size_t)np * PRP_INTS * sizeof(int), hipMemcpyDeviceToDevice, gs));
    sa.prp_given = d_prp_src ? 1 : 0;
    sa.order_flag = order_flag; sa.order_gen = order_gen; sa.detect_order = (dim == 8 && c->order_mode == LT_ORDER_AUTO) ? 1 : 0;
    // object batches: unclipped triangles take the fast kernel, the rest go through sub-queues
    sa.queue = screen ? nullptr : wa.queue;
    sa.queue_count = (unsigned *)(order_flag + 64);      // behind the order flag
    sa.queue_cap = pl.qcap;
    sa.queue_index_limit = (unsigned)np;
    sa.count_drawn = 1;
    sa.sq_desc = wa.sq_desc; sa.sq_tuples = wa.sq_tuples; sa.sq_queue = wa.sq_queue;
    sa.sq_count = sa.queue_count + QUEUES * QUEUE_COUNT_STRIDE;
    sa.band_y0 = c->band_y0; sa.band_y1 = c->band_y1;
    sa.cx0 = c->clip.x0; sa.cy0 = c->clip.y0; sa.cx1 = c->clip.x1; sa.cy1 = c->clip.y1;
    // Screens with more tiles than the LDS histogram of the count kernel holds: set-up adds the
    // references itself (with few tiles the adds pile up on the same words and stall set-up).
    const bool fused = ntiles > BIN_MAX_TILES;
    sa.tile_count = fused ? wa.tcount : nullptr;
    sa.status = c->status;
    launch_setup(gs, dim, screen, sa);
    if ((rc = stage_end(c, 1, gs))) return rc;

    // count + scan are the "scan" stage, the scatter is the "bin" stage; the event between
    // them is recorded inside launch_binning
    if ((rc = stage_begin(c, 2, gs))) return rc;
    hipEvent_t scan_ev = nullptr;
    size_t scan_idx = 0;
    if (timed(c, 2) || timed(c, 3)) {
        if ((rc = new_event(c, (hipStream_t)-1, &scan_idx))) return rc;
        scan_ev = c->ev_pool[scan_idx];
    }
    const bool ordered_follows = dim == 8 && c->order_mode != LT_ORDER_NEVER;
    const bool band_lists = ordered_follows && wa.blist;
    const int order_force = (dim == 8 && c->order_mode == LT_ORDER_ALWAYS) ? 1 : 0;
    launch_binning(gs, wa.hdr, np, twl, tiles_x, ntiles, wa.tcount, wa.toffset, wa.tcursor, wa.list, capacity,
                   screen ? nullptr : sa.queue_count, c->status, scan_ev, fused, geom_done,
                   ordered_follows ? order_flag : nullptr, order_gen, order_force);
    if ((rc = stage_end(c, 2, gs, &scan_idx))) return rc;
    if (timed(c, 3)) {
        lt_ctx::Span sp{3, scan_idx, 0};
        c->spans.push_back(sp);
        if ((rc = stage_end(c, 3, gs))) return rc;
    }
    BandArgs ba{};
    if (band_lists) {
        ba.hdr = wa.hdr; ba.np = np;
        ba.R = pl.band_rows; ba.nb = pl.nbands;
        ba.cnt = wa.bcnt; ba.base = wa.bplace; ba.total = wa.bplace + pl.nbands;
        ba.list = wa.blist; ba.capacity = pl.band_capacity;
        ba.alloc = (unsigned *)(order_flag + 1);
        ba.order_flag = order_flag; ba.order_gen = order_gen; ba.force_order = order_force;
        ba.status = c->status;
    }
    if (geom_done) HIPCHK(c, hipStreamWaitEvent(rs, geom_done, 0));      // the scatter's completion

    RasterArgs ra{};
    ra.hdr = wa.hdr; ra.edges = wa.edges;
    ra.tile_offset = wa.toffset; ra.tile_fill = wa.tcursor;
    ra.tile_list = wa.list; ra.list_capacity = capacity;
    ra.np = np; ra.emax = emax;
    ra.W = c->W; ra.H = c->H; ra.tiles_x = tiles_x; ra.tiles_y = c->tiles_y;
    ra.cbuf = c->cbuf; ra.zbuf = c->zbuf;
    ra.clear_pending = c->clear_pending ? 1 : 0; ra.clear_colour = c->clear_colour;
    ra.prp = want_prp ? wa.prp : nullptr; ra.log_focus = log_focus;
    ra.tex_ptr = (const int *const *)c->tex_ptr_d; ra.tex_log = c->tex_log_d;
    ra.order_flag = order_flag; ra.order_gen = order_gen;
    ra.force_order = order_force;
    ra.band_y0 = c->band_y0; ra.band_y1 = c->band_y1;
    ra.status = c->status;
    // A Gouraud or flat fill that shares the chip with the set-up chains of other batches leaves
    // them room in LDS (not the textured fill: measured slower with it, 1.148 -> 1.159 ms at 4K)
    const int waves = tile_w == TW_NARROW && (c->tile_waves ? c->tile_waves == 8 : c->auto_waves8) ? 8 : 4;
    const unsigned fill_lds = (geom_done && c->fill_shares_chip && tile_w == TW_NARROW && dim != 8 && waves == 4) ? FILL_OVERLAP_LDS : 0u;
    // The fill's events are its own dispatch's where no other kernel follows it: its start
    // and end for the stage timing, else its completion as this work set's "free" signal.
    if (ordered_follows) {
        if ((rc = stage_begin(c, 4, rs))) return rc;
        launch_raster(rs, na, tile_w, ra, nullptr, nullptr, fill_lds, waves);
        // (the per-band lists on the fill's stream: made on the geometry stream, beside the fill of
        // the batch before, they slow that fill down by more than they take here)
        if (band_lists) HIPCHK(c, launch_band_lists(rs, np, c->device, ba));
        HIPCHK(c, launch_ordered(rs, ra, c->device, band_lists ? &ba : nullptr));
        if ((rc = stage_end(c, 4, rs))) return rc;
        if (raster_done) HIPCHK(c, hipEventRecord(raster_done, rs));
    } else if (timed(c, 4)) {
        lt_ctx::Span sp{4, 0, 0};
        if ((rc = new_event(c, (hipStream_t)-1, &sp.begin))) return rc;
        if ((rc = new_event(c, (hipStream_t)-1, &sp.end))) return rc;
        c->spans.push_back(sp);
        launch_raster(rs, na, tile_w, ra, c->ev_pool[sp.begin], c->ev_pool[sp.end], fill_lds, waves);
        if (raster_done) HIPCHK(c, hipEventRecord(raster_done, rs));
    } else {
        launch_raster(rs, na, tile_w, ra, nullptr, raster_done, fill_lds, waves);
    }
    HIPCHK(c, hipGetLastError());
    c->batch_no++;
    if (timing && c->profiling && (c->batch_no - 1) % c->profile_every == 0) c->timed_batches++;

    c->clear_pending = false;
    c->frame_defined = true;
    c->drawn_since_clear = true;
    c->polys_in += (uint64_t)np;
    return LT_OK;
}

// One batch through a work set of the context (grown on demand).
int run_batch(lt_ctx *c, bool screen, int dim, const int32_t *m, const int *d_verts, int nv, const int *d_polys, const int *d_aux,
              int np, const int *d_pv, int npv, int render_type, int log_focus, int force_linear, int max_edges,
              const int *d_prp_src = nullptr, const ObjectDesc *d_objects = nullptr, const int *d_polyobj = nullptr, int nobj = 0)
{
    const BatchPlan pl = plan_batch(c, screen, dim, np, max_edges, 1u << 20, d_prp_src != nullptr);
    int rc;
    // batches whose work arrays stay small take four sets and both geometry streams in turns
    const bool deep = c->pipelined && np <= (1 << 24) && pl.capacity <= (1ull << 30);
    const int set = (int)(c->batch_no % (deep ? 4u : 2u));
    lt_ctx::Work &w = c->work[set];
    hipStream_t gs = (deep && (c->batch_no & 1u)) ? c->own_gs2 : c->gs, rs = c->rs;
    if (c->pipelined) {
        // the inputs may have been produced on the caller's stream; an idle stream has
        // nothing to wait for, and every wait costs the set-up stream a few microseconds
        if (hipStreamQuery(c->stream) != hipSuccess) {
            (void)hipGetLastError();        // "not ready" is an answer, not an error
            HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
            HIPCHK(c, hipStreamWaitEvent(gs, c->ev_fork, 0));
        }
        // this workspace set is free once the fill of the batch two before this one is done
        if (w.in_use) HIPCHK(c, hipStreamWaitEvent(gs, w.raster_done, 0));
        // Fragment-heavy batches gain nothing from overlap: set-up is a twentieth of the fill
        // and the two slow each other down (measured: 1.79 ms per frame one after the other,
        // 2.3-2.5 ms overlapped), so there the set-up waits for the fill before it as well.
        lt_ctx::Work &prev = c->work[c->last_set];
        if (c->fragment_heavy && prev.in_use) HIPCHK(c, hipStreamWaitEvent(gs, prev.raster_done, 0));
        // Is the fill of the batch before still on its way?  Then batches arrive faster than they are
        // drawn, and while this one's fill runs the set-up of the next will want room beside it
        // (FILL_OVERLAP_LDS).  A caller that waits for every frame gets the fill at full occupancy.
        c->fill_shares_chip = false;
        if (prev.in_use && !c->fragment_heavy) {
            c->fill_shares_chip = hipEventQuery(prev.raster_done) == hipErrorNotReady;
            (void)hipGetLastError();        // "not ready" is an answer, not an error
        }
    } else {
        c->fill_shares_chip = false;
    }
    if ((rc = ensure(c, w.hdr, pl.hdr))) return rc;
    if ((rc = ensure(c, w.edges, pl.edges))) return rc;
    if ((rc = ensure(c, w.list, pl.list))) return rc;
    if (pl.want_prp && (rc = ensure(c, w.prp, pl.prp))) return rc;
    // the tile counts and the sub-queue counters are zero: allocated so, and the scan kernel
    // of the batch that used them last has reset them
    if ((rc = ensure_zeroed(c, w.tcount, pl.tcount, gs))) return rc;
    if ((rc = ensure(c, w.toffset, pl.toffset))) return rc;
    if ((rc = ensure(c, w.tcursor, pl.tcursor))) return rc;
    if ((rc = ensure_zeroed(c, w.scratch, pl.flags, gs))) return rc;
    if (!screen && (rc = ensure(c, w.queue, pl.queue))) return rc;
    if (!screen && (rc = ensure(c, w.sq_queue, pl.sq_queue))) return rc;
    if (!screen && (rc = ensure(c, w.sq_desc, pl.sq_desc))) return rc;
    if (!screen && (rc = ensure(c, w.sq_tuples, pl.sq_tuples))) return rc;
    if (pl.nbands) {
        if ((rc = ensure(c, w.bcnt, pl.bcnt))) return rc;
        if ((rc = ensure(c, w.bplace, pl.bplace))) return rc;
        if ((rc = ensure(c, w.blist, pl.blist))) return rc;
    }
    WorkArrays wa;
    wa.hdr = (PrimHdr *)w.hdr.p; wa.edges = (int *)w.edges.p; wa.prp = (int *)w.prp.p;
    wa.list = (unsigned *)w.list.p; wa.tcount = (unsigned *)w.tcount.p; wa.toffset = (unsigned *)w.toffset.p;
    wa.tcursor = (unsigned *)w.tcursor.p; wa.flags = (int *)w.scratch.p;
    wa.queue = (uint2 *)w.queue.p; wa.sq_queue = (uint2 *)w.sq_queue.p;
    wa.sq_desc = (int *)w.sq_desc.p; wa.sq_tuples = (int *)w.sq_tuples.p;
    if (pl.nbands) { wa.bcnt = (unsigned *)w.bcnt.p; wa.bplace = (unsigned *)w.bplace.p; wa.blist = (unsigned *)w.blist.p; }
    rc = issue_batch(c, wa, pl, gs, rs, c->pipelined ? w.geom_done : nullptr, c->pipelined ? w.raster_done : nullptr, true,
                     screen, dim, m, d_verts, nv, d_polys, d_aux, np, d_pv, npv, render_type, log_focus, force_linear, d_prp_src,
                     d_objects, d_polyobj, nobj);
    if (rc) return rc;
    if (c->pipelined) w.in_use = true;
    c->last_set = set;
    return LT_OK;
}

// G_clear with no fill kernel to do it: the rows of this context's band
void fill_band(lt_ctx *c)
{
    if (!c->clear_pending) return;
    const int y0 = c->band_y0 < 0 ? 0 : c->band_y0, y1 = c->band_y1 > c->H ? c->H : c->band_y1;
    if (y0 < y1) launch_fill(c->rs, c->cbuf + (size_t)y0 * c->W, c->zbuf + (size_t)y0 * c->W, (size_t)(y1 - y0) * c->W, c->clear_colour);
    c->clear_pending = false;
}

// Wireframe batches (M_WIRE objects, G_line lists) read the depth buffer and never write it, so
// they run where the fills run, after them, and the next fill runs after them.
int wire_begin(lt_ctx *c, hipStream_t *s)
{
    *s = c->rs;
    c->drawn_since_clear = true;
    if (c->alt_cbuf && c->pipelined) {
        // the per-pixel winner words are shared by both frames in flight: lines wait for
        // whatever the other frame's stream still holds
        hipStream_t other = c->rs == c->alt_rs ? c->own_rs : c->alt_rs;
        HIPCHK(c, hipEventRecord(c->ev_side, other));
        HIPCHK(c, hipStreamWaitEvent(c->rs, c->ev_side, 0));
    }
    if (c->pipelined && hipStreamQuery(c->stream) != hipSuccess) {
        (void)hipGetLastError();
        HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));       // the inputs may come from the caller's stream
        HIPCHK(c, hipStreamWaitEvent(c->rs, c->ev_fork, 0));
    }
    fill_band(c);
    const size_t bytes = (size_t)c->W * c->H * sizeof(unsigned);
    if (bytes > c->line_winner.cap) {
        int rc = ensure(c, c->line_winner, bytes);
        if (rc) return rc;
        HIPCHK(c, hipMemsetAsync(c->line_winner.p, 0, c->line_winner.cap, c->rs));     // the resolve kernel zeroes what it reads
    }
    return LT_OK;
}

int run_wire(lt_ctx *c, const int32_t *m, const int *d_verts, int nv, const int *d_polys, const int *d_aux, int np,
             const int *d_pv, int npv, int log_focus, int max_edges)
{
    if (np > (1 << 27)) return fail(c, LT_ERR_ARG, "lt_submit_object: at most %d polygons in a wireframe batch", 1 << 27);
    hipStream_t s;
    int rc = wire_begin(c, &s);
    if (rc) return rc;
    if ((rc = ensure(c, c->line_colours, (size_t)np * sizeof(int)))) return rc;
    SetupArgs sa{};
    sa.use_matrix = m ? 1 : 0;
    if (m) for (int i = 0; i < 12; i++) sa.m[i] = m[i];
    sa.verts = d_verts; sa.polys = d_polys; sa.polyaux = d_aux; sa.polyverts = d_pv;
    sa.np = np; sa.nv = nv; sa.npv = npv;
    sa.log_focus = log_focus;
    sa.max_edges = (max_edges <= 0 || max_edges > 15) ? 15 : max_edges;
    sa.W = c->W; sa.H = c->H;
    sa.band_y0 = c->band_y0; sa.band_y1 = c->band_y1;
    sa.cx0 = c->clip.x0; sa.cy0 = c->clip.y0; sa.cx1 = c->clip.x1; sa.cy1 = c->clip.y1;
    sa.status = c->status;
    launch_wire(s, sa, c->zbuf, (unsigned *)c->line_winner.p, (int *)c->line_colours.p, c->cbuf);
    HIPCHK(c, hipGetLastError());
    c->frame_defined = true;
    c->polys_in += (uint64_t)np;
    return LT_OK;
}

int check_type(lt_ctx *c, int render_type, int *dim)
{
    if (render_type == LT_FLAT) *dim = 3;
    else if (render_type == LT_SHADED) *dim = 6;
    else if (render_type == LT_TEXTURED) *dim = 8;
    else return fail(c, LT_ERR_ARG, "render type %d: not LT_FLAT, LT_SHADED or LT_TEXTURED (LT_WIRE is for lt_submit_object only)", render_type);
    return LT_OK;
}

}  // namespace

extern "C" {

int lt_ctx_create(int device, int width, int height, lt_ctx **out)
{
    if (!out) return fail(nullptr, LT_ERR_ARG, "lt_ctx_create: out is NULL");
    *out = nullptr;
    if (width < 2 || height < 2 || width > LT_MAX_SCREEN || height > LT_MAX_SCREEN)
        return fail(nullptr, LT_ERR_ARG, "lt_ctx_create: screen %dx%d outside 2..%d", width, height, LT_MAX_SCREEN);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, LT_ERR_HIP, "lt_ctx_create: no HIP device (%s)", hipGetErrorString(e));
    if (device < 0 || device >= ndev || device >= MAX_DEVICES) return fail(nullptr, LT_ERR_ARG, "lt_ctx_create: device %d of %d", device, ndev);
    lt_ctx *c = new lt_ctx();
    c->device = device;
    c->W = width; c->H = height;
    c->band_y0 = 0; c->band_y1 = height;
    c->clip = ClipWindow{0, 0, width - 1, height - 1};     // C_init_clipping defaults, Clipper.h:19-23
    c->tiles_x = (width + TW_NARROW - 1) / TW_NARROW; c->tiles_y = (height + TH - 1) / TH;
    c->ntiles = c->tiles_x * c->tiles_y;
    size_t n = (size_t)width * height;
#define CREATECHK(call)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            int rc_ = fail(nullptr, LT_ERR_HIP, "lt_ctx_create: %s: %s", #call, hipGetErrorString(e_));   \
            lt_ctx_destroy(c);                                                                            \
            return rc_;                                                                                   \
        }                                                                                                 \
    } while (0)
    CREATECHK(hipSetDevice(device));
    CREATECHK(noalloc_init(device));
    CREATECHK(hipMalloc((void **)&c->own_cbuf, n * sizeof(int)));
    CREATECHK(hipMalloc((void **)&c->own_zbuf, n * sizeof(int)));
    c->cbuf = c->own_cbuf; c->zbuf = c->own_zbuf;
    CREATECHK(hipMalloc((void **)&c->status, sizeof(Status) * STATUS_SLOTS));
    CREATECHK(hipMemset(c->status, 0, sizeof(Status) * STATUS_SLOTS));
    CREATECHK(hipStreamCreateWithFlags(&c->own_gs, hipStreamNonBlocking));
    CREATECHK(hipStreamCreateWithFlags(&c->own_rs, hipStreamNonBlocking));
    CREATECHK(hipStreamCreateWithFlags(&c->alt_rs, hipStreamNonBlocking));
    CREATECHK(hipStreamCreateWithFlags(&c->own_gs2, hipStreamNonBlocking));
    c->gs = c->own_gs; c->rs = c->own_rs;
    // Two frames in flight keep four streams of the library busy beside the caller's.  HIP gives a
    // process GPU_MAX_HW_QUEUES hardware queues (four unless the variable says otherwise, read when
    // HIP starts) and lets streams share them; with four, two of these streams share one and the
    // second frame costs more than it brings (0.169 against 0.163 ms per C2-small frame), so the
    // default follows the variable.  lt_set_frame_overlap overrides it either way.
    {
        const char *q = getenv("GPU_MAX_HW_QUEUES");
        c->frame_overlap = q && atoi(q) >= 8;
    }
    CREATECHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    CREATECHK(hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming));
    for (lt_ctx::Work &w : c->work) {
        CREATECHK(hipEventCreateWithFlags(&w.geom_done, hipEventDisableTiming));
        CREATECHK(hipEventCreateWithFlags(&w.raster_done, hipEventDisableTiming));
    }
    CREATECHK(hipMalloc((void **)&c->tex_ptr_d, sizeof(int *) * LT_MAX_TEXTURES));
    CREATECHK(hipMalloc((void **)&c->tex_log_d, sizeof(int) * LT_MAX_TEXTURES));
    CREATECHK(hipMemset((void *)c->tex_ptr_d, 0, sizeof(int *) * LT_MAX_TEXTURES));
    CREATECHK(hipMemset(c->tex_log_d, 0, sizeof(int) * LT_MAX_TEXTURES));
    // the fills above ran on the default stream, which the context's own streams do not wait for
    CREATECHK(hipDeviceSynchronize());
#undef CREATECHK
    // the frame's contents are undefined until lt_clear, as after G_init_graphics
    *out = c;
    return LT_OK;
}

int lt_ctx_destroy(lt_ctx *c)
{
    if (!c) return LT_OK;
    (void)hipSetDevice(c->device);
    if (c->gs) (void)hipStreamSynchronize(c->gs);
    if (c->own_gs2) (void)hipStreamSynchronize(c->own_gs2);
    if (c->rs) (void)hipStreamSynchronize(c->rs);
    if (c->own_rs) (void)hipStreamSynchronize(c->own_rs);
    if (c->alt_rs) (void)hipStreamSynchronize(c->alt_rs);
    (void)hipStreamSynchronize(c->stream);
    c->host_stage.stop();
    if (c->host_stage.ring) (void)hipHostFree(c->host_stage.ring);
    for (hipEvent_t e : c->host_stage.sent)
        if (e) (void)hipEventDestroy(e);
    DevBuf *bufs[] = {&c->s_verts, &c->s_polys, &c->s_aux, &c->s_pv, &c->s_normals, &c->s_pn, &c->s_prp, &c->s_objects, &c->s_polyobj,
                      &c->line_winner, &c->line_colours, &c->s_lines};
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (lt_ctx::Work &w : c->work) {
        DevBuf *wb[] = {&w.hdr, &w.edges, &w.prp, &w.list, &w.tcount, &w.toffset, &w.tcursor, &w.queue, &w.scratch,
                        &w.sq_desc, &w.sq_tuples, &w.sq_queue, &w.bcnt, &w.bplace, &w.blist};
        for (DevBuf *b : wb)
            if (b->p) (void)hipFree(b->p);
        if (w.geom_done) (void)hipEventDestroy(w.geom_done);
        if (w.raster_done) (void)hipEventDestroy(w.raster_done);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->own_gs) (void)hipStreamDestroy(c->own_gs);
    if (c->own_gs2) (void)hipStreamDestroy(c->own_gs2);
    if (c->own_rs) (void)hipStreamDestroy(c->own_rs);
    if (c->alt_rs) (void)hipStreamDestroy(c->alt_rs);
    if (c->ev_side) (void)hipEventDestroy(c->ev_side);
    if (c->alt_cbuf) (void)hipFree(c->alt_cbuf);
    if (c->alt_zbuf) (void)hipFree(c->alt_zbuf);
    for (int i = 0; i < LT_MAX_TEXTURES; i++)
        if (c->tex_data[i]) (void)hipFree(c->tex_data[i]);
    if (c->own_cbuf) (void)hipFree(c->own_cbuf);
    if (c->own_zbuf) (void)hipFree(c->own_zbuf);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->status) (void)hipFree(c->status);
    if (c->tex_ptr_d) (void)hipFree((void *)c->tex_ptr_d);
    if (c->tex_log_d) (void)hipFree(c->tex_log_d);
    delete c;
    return LT_OK;
}

const char *lt_last_error(lt_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int lt_set_stream(lt_ctx *c, void *hip_stream)
{
    if (!c) return LT_ERR_ARG;
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = sync_all(c))) return rc;
    c->stream = (hipStream_t)hip_stream;
    if (!c->pipelined) c->gs = c->rs = c->stream;
    return LT_OK;
}

int lt_set_pipelining(lt_ctx *c, int on)
{
    if (!c) return LT_ERR_ARG;
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = sync_all(c))) return rc;
    c->pipelined = on != 0;
    c->gs = c->pipelined ? c->own_gs : c->stream;
    c->rs = !c->pipelined ? c->stream : (c->cbuf == c->alt_cbuf && c->alt_rs ? c->alt_rs : c->own_rs);
    for (lt_ctx::Work &w : c->work) w.in_use = false;
    return LT_OK;
}

int lt_set_tile_width(lt_ctx *c, int width)
{
    if (!c || (width != 0 && width != TW_NARROW && width != TW_WIDE)) return fail(c, LT_ERR_ARG, "lt_set_tile_width: 0, %d or %d", TW_NARROW, TW_WIDE);
    c->tile_mode = width;
    return LT_OK;
}

int lt_set_tile_waves(lt_ctx *c, int waves)
{
    if (!c || (waves != 0 && waves != 4 && waves != 8)) return fail(c, LT_ERR_ARG, "lt_set_tile_waves: 0, 4 or 8");
    c->tile_waves = waves;
    return LT_OK;
}

int lt_set_copy_threads(lt_ctx *c, int threads)
{
    if (!c || threads < 1 || threads > HostStage::MAX_HELPERS + 1) return fail(c, LT_ERR_ARG, "lt_set_copy_threads: 1..%d", HostStage::MAX_HELPERS + 1);
    c->host_stage.stop();                   // the helpers start again, in the new number, with the next large host array
    c->host_stage.HELPERS = threads - 1;
    return LT_OK;
}

int lt_host_alloc(size_t bytes, void **out)
{
    if (!out || bytes == 0) return LT_ERR_ARG;
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocPortable);
    if (e != hipSuccess) {
        *out = nullptr;
        return fail(nullptr, LT_ERR_HIP, "lt_host_alloc: %zu bytes: %s", bytes, hipGetErrorString(e));
    }
    return LT_OK;
}

int lt_host_free(void *p)
{
    if (!p) return LT_OK;
    hipError_t e = hipHostFree(p);
    return e == hipSuccess ? LT_OK : fail(nullptr, LT_ERR_HIP, "lt_host_free: %s", hipGetErrorString(e));
}

int lt_set_textured_order(lt_ctx *c, int mode)
{
    const int lists = mode & (LT_ORDER_LISTS_ALWAYS | LT_ORDER_LISTS_NEVER);
    mode &= ~(LT_ORDER_LISTS_ALWAYS | LT_ORDER_LISTS_NEVER);
    if (!c || mode < LT_ORDER_AUTO || mode > LT_ORDER_NEVER || lists == (LT_ORDER_LISTS_ALWAYS | LT_ORDER_LISTS_NEVER))
        return fail(c, LT_ERR_ARG, "lt_set_textured_order: mode %d", mode | lists);
    c->order_mode = mode;
    c->order_lists = lists;
    return LT_OK;
}

int lt_set_list_capacity(lt_ctx *c, uint64_t refs)
{
    if (!c) return LT_ERR_ARG;
    c->list_capacity_user = refs;
    return LT_OK;
}

int lt_set_band(lt_ctx *c, int y0, int y1)
{
    if (!c) return LT_ERR_ARG;
    if (y0 < 0 || y1 > c->H || y0 >= y1) return fail(c, LT_ERR_ARG, "lt_set_band: rows [%d, %d) outside [0, %d)", y0, y1, c->H);
    c->band_y0 = y0; c->band_y1 = y1;
    return LT_OK;
}

int lt_set_clip_window(lt_ctx *c, int minx, int miny, int maxx, int maxy)
{
    if (!c) return LT_ERR_ARG;
    if (minx < 0 || miny < 0 || maxx > c->W - 1 || maxy > c->H - 1 || minx >= maxx || miny >= maxy)
        return fail(c, LT_ERR_ARG, "lt_set_clip_window: (%d, %d)-(%d, %d) is empty or leaves the %d x %d screen", minx, miny, maxx,
                    maxy, c->W, c->H);
    c->clip = ClipWindow{minx, miny, maxx, maxy};
    return LT_OK;
}

int lt_bind_framebuffer(lt_ctx *c, int32_t *d_colour, int32_t *d_depth)
{
    if (!c) return LT_ERR_ARG;
    if ((d_colour == nullptr) != (d_depth == nullptr)) return fail(c, LT_ERR_ARG, "lt_bind_framebuffer: give both arrays or neither");
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = lt_flush(c))) return rc;            // queued work and a pending clear go to the frame they were issued for
    // unbinding returns to the own frame as it was left, in whichever of the two pairs it is
    c->cbuf = d_colour ? d_colour : (c->on_alt ? c->alt_cbuf : c->own_cbuf);
    c->zbuf = d_depth ? d_depth : (c->on_alt ? c->alt_zbuf : c->own_zbuf);
    if (c->pipelined) c->rs = (!d_colour && c->on_alt) ? c->alt_rs : c->own_rs;
    c->frame_defined = true;                      // the memory holds whatever its owner left there
    return LT_OK;
}

int lt_set_frame_overlap(lt_ctx *c, int on)
{
    if (!c) return LT_ERR_ARG;
    c->frame_overlap = on != 0;
    return LT_OK;
}

// the frame that starts with this clear goes to the other buffer pair and fill stream
static int next_frame_side(lt_ctx *c)
{
    const bool own = c->cbuf == c->own_cbuf || (c->alt_cbuf && c->cbuf == c->alt_cbuf);
    if (!c->frame_overlap || !c->pipelined || c->fb_exposed || !own || !c->drawn_since_clear || c->band_y0 > 0 || c->band_y1 < c->H)
        return LT_OK;
    int rc = use_device(c);
    if (rc) return rc;
    if (!c->alt_cbuf) {
        const size_t n = (size_t)c->W * c->H;
        HIPCHK(c, hipMalloc((void **)&c->alt_cbuf, n * sizeof(int)));
        HIPCHK(c, hipMalloc((void **)&c->alt_zbuf, n * sizeof(int)));
    }
    const bool to_alt = c->cbuf == c->own_cbuf;
    c->on_alt = to_alt;
    c->cbuf = to_alt ? c->alt_cbuf : c->own_cbuf;
    c->zbuf = to_alt ? c->alt_zbuf : c->own_zbuf;
    c->rs = to_alt ? c->alt_rs : c->own_rs;
    return LT_OK;
}

int lt_clear(lt_ctx *c, int32_t lit_colour)
{
    if (!c) return LT_ERR_ARG;
    int rc = next_frame_side(c);
    if (rc) return rc;
    c->drawn_since_clear = false;
    c->clear_pending = true;
    c->clear_colour = lit_colour;
    c->frame_defined = true;
    return LT_OK;
}

static int materialize_clear(lt_ctx *c)
{
    fill_band(c);
    HIPCHK(c, hipGetLastError());
    return LT_OK;
}

int lt_flush(lt_ctx *c)
{
    if (!c) return LT_ERR_ARG;
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = materialize_clear(c))) return rc;
    if ((rc = sync_all(c))) return rc;
    if ((rc = collect(c))) return rc;
    Status s;
    if ((rc = read_status(c, &s))) return rc;
    // Tile width of the batches to come: wide once the polygons drawn since the last look
    // averaged 256 depth tests or more (a span is walked once per tile it crosses).
    if (s.polys_drawn >= c->seen_drawn + 256) {
        c->auto_wide = (s.fragments - c->seen_fragments) / (s.polys_drawn - c->seen_drawn) >= 256;
        c->fragment_heavy = c->auto_wide;
        c->seen_fragments = s.fragments; c->seen_drawn = s.polys_drawn;
    }
    // Waves on a 64-pixel tile for the batches to come: eight once the longest tile list of the batches since
    // the last look averaged 2048 entries or more and eight times the average list (polygons
    // crowded into few tiles: the fill lasts as long as its longest list).
    if (c->batch_no > c->seen_batches) {
        const unsigned long long nb = c->batch_no - c->seen_batches;
        const unsigned long long longest = (s.longest_lists - c->seen_longest) / nb;
        const unsigned long long mean = (s.tile_refs - c->seen_refs) / nb / (unsigned long long)(c->ntiles > 0 ? c->ntiles : 1);
        c->auto_waves8 = longest >= 2048 && longest >= 8 * (mean > 0 ? mean : 1);
        c->seen_longest = s.longest_lists; c->seen_refs = s.tile_refs; c->seen_batches = c->batch_no;
    }
    if (s.list_overflow > c->base.list_overflow) {
        c->base.list_overflow = s.list_overflow;
        return fail(c, LT_ERR_CAPACITY,
                    "tile lists overflowed by %llu references: the frame is incomplete; raise lt_set_list_capacity",
                    (unsigned long long)s.list_overflow);
    }
    if (s.overflow_polys > c->base.overflow_polys) {
        unsigned long long d = s.overflow_polys - c->base.overflow_polys;
        c->base.overflow_polys = s.overflow_polys;
        return fail(c, LT_ERR_OVERFLOW,
                    "%llu polygon(s) outside the limits (fewer than 2 or more than max_edges edges, a vertex index "
                    "out of range, or more than 16 tuples after clipping) were not drawn", d);
    }
    if (s.unsupported_polys > c->base.unsupported_polys) {
        unsigned long long d = s.unsupported_polys - c->base.unsupported_polys;
        c->base.unsupported_polys = s.unsupported_polys;
        return fail(c, LT_ERR_UNSUPPORTED,
                    "%llu textured polygon(s) name a bad texture, or take the perspective mapper with a singular "
                    "texture mapping (the reference divides by zero there) or without polyaux: not drawn", d);
    }
    if (s.zero_light_vectors > c->base.zero_light_vectors) {
        unsigned long long d = s.zero_light_vectors - c->base.zero_light_vectors;
        c->base.zero_light_vectors = s.zero_light_vectors;
        return fail(c, LT_ERR_UNSUPPORTED,
                    "%llu zero-length vector(s) -- a point light at a vertex, or a polygon whose first three vertices "
                    "span no plane: the reference stops there with 'Found a zero length vector'; that light was left "
                    "out for that vertex, that plane normal is {0,0,0}", d);
    }
    if (s.order_dependent_rows > c->base.order_dependent_rows) {
        unsigned long long d = s.order_dependent_rows - c->base.order_dependent_rows;
        c->base.order_dependent_rows = s.order_dependent_rows;
        return fail(c, LT_ERR_UNSUPPORTED,
                    "%llu textured pixel(s) lie on rows with a light gradient while LT_ORDER_NEVER is set: the "
                    "reference advances light only on depth-test passes there; drawn as if every earlier pixel "
                    "had passed", d);
    }
    return LT_OK;
}

int lt_readback(lt_ctx *c, int32_t *colour, int32_t *depth)
{
    if (!c) return LT_ERR_ARG;
    if (!c->frame_defined) return fail(c, LT_ERR_ARG, "lt_readback before the first lt_clear: the frame is undefined");
    int rc = lt_flush(c);
    size_t bytes = (size_t)c->W * c->H * sizeof(int);
    int rc2 = LT_OK;
    if (colour && (rc2 = d2h(c, colour, c->cbuf, bytes))) return rc2;
    if (depth && (rc2 = d2h(c, depth, c->zbuf, bytes))) return rc2;
    return rc;
}

int lt_framebuffer(lt_ctx *c, int32_t **d_colour, int32_t **d_depth)
{
    if (!c) return LT_ERR_ARG;
    c->fb_exposed = true;             // the caller may keep the addresses: the frame stays where it is from now on
    if (d_colour) *d_colour = c->cbuf;
    if (d_depth) *d_depth = c->zbuf;
    return LT_OK;
}

int lt_get_counters(lt_ctx *c, lt_counters *out)
{
    if (!c || !out) return LT_ERR_ARG;
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = sync_all(c))) return rc;
    Status s;
    if ((rc = read_status(c, &s))) return rc;
    out->fragments = s.fragments;
    out->polys_in = c->polys_in;
    out->polys_drawn = s.polys_drawn;
    out->tile_refs = s.tile_refs;
    out->overflow_polys = s.overflow_polys;
    out->order_dependent_rows = s.order_dependent_rows;
    out->unsupported_polys = s.unsupported_polys;
    out->list_overflow = s.list_overflow;
    out->zero_light_vectors = s.zero_light_vectors;
    return LT_OK;
}

int lt_reset_counters(lt_ctx *c)
{
    if (!c) return LT_ERR_ARG;
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = sync_all(c))) return rc;
    HIPCHK(c, hipMemsetAsync(c->status, 0, sizeof(Status) * STATUS_SLOTS, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->base = Status{};
    c->polys_in = 0;
    c->seen_fragments = 0; c->seen_drawn = 0;
    c->seen_longest = 0; c->seen_refs = 0; c->seen_batches = c->batch_no;
    return LT_OK;
}

int lt_set_profiling(lt_ctx *c, int on)
{
    if (!c) return LT_ERR_ARG;
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = sync_all(c))) return rc;
    if ((rc = collect(c))) return rc;
    c->profiling = on == 1 ? (1 << LT_STAGES) - 1 : (on & ((1 << LT_STAGES) - 1));
    c->profile_every = on > 1 && (on >> 8) > 1 ? (unsigned)(on >> 8) : 1u;
    return LT_OK;
}

int lt_get_stage_ms(lt_ctx *c, double ms[LT_STAGES], int *submits)
{
    if (!c || !ms) return LT_ERR_ARG;
    int rc = lt_flush(c);
    for (int k = 0; k < LT_STAGES; k++) { ms[k] = c->stage_ms[k]; c->stage_ms[k] = 0; }
    if (submits) *submits = c->stage_submits;
    c->stage_submits = 0;
    return rc;
}

int lt_texture_upload(lt_ctx *c, int handle, const int32_t *texels, int log_size)
{
    if (!c || !texels) return LT_ERR_ARG;
    if (handle < 0 || handle >= LT_MAX_TEXTURES) return fail(c, LT_ERR_ARG, "texture handle %d outside 0..%d", handle, LT_MAX_TEXTURES - 1);
    if (log_size < 0 || log_size > 12) return fail(c, LT_ERR_ARG, "texture log size %d outside 0..12", log_size);
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = sync_all(c))) return rc;
    if (c->tex_data[handle]) { HIPCHK(c, hipFree(c->tex_data[handle])); c->tex_data[handle] = nullptr; }
    size_t bytes = sizeof(int) << (2 * log_size);
    HIPCHK(c, hipMalloc((void **)&c->tex_data[handle], bytes));
    HIPCHK(c, hipMemcpy(c->tex_data[handle], texels, bytes, hipMemcpyHostToDevice));
    c->tex_log_h[handle] = log_size;
    HIPCHK(c, hipMemcpy((void *)(c->tex_ptr_d + handle), &c->tex_data[handle], sizeof(int *), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->tex_log_d + handle, &log_size, sizeof(int), hipMemcpyHostToDevice));
    return LT_OK;
}

int lt_transform(lt_ctx *c, const int32_t m[12], const int32_t *d_from, int32_t *d_to, int n)
{
    if (!c || !m || n < 0 || (n > 0 && (!d_from || !d_to))) return fail(c, LT_ERR_ARG, "lt_transform: bad argument");
    int rc = use_device(c);
    if (rc) return rc;
    launch_transform(c->stream, m, d_from, d_to, n);
    HIPCHK(c, hipGetLastError());
    return LT_OK;
}

int lt_polygon_rows_dev(lt_ctx *c, const int32_t m[12], const int32_t *d_verts, int nv, const int32_t *d_polys, int np,
                        const int32_t *d_polyverts, int npv, int log_focus, int max_edges, int32_t *d_rows)
{
    if (!c) return LT_ERR_ARG;
    if (np < 0 || nv < 0 || npv < 0) return fail(c, LT_ERR_ARG, "lt_polygon_rows_dev: negative count");
    if (log_focus < 0 || log_focus > 30) return fail(c, LT_ERR_ARG, "lt_polygon_rows_dev: log_focus %d outside 0..30", log_focus);
    if (max_edges <= 0 || max_edges > 15) max_edges = 15;
    if (np == 0) return LT_OK;
    if (!d_verts || !d_polys || !d_polyverts || !d_rows) return fail(c, LT_ERR_ARG, "lt_polygon_rows_dev: NULL array");
    int rc = use_device(c);
    if (rc) return rc;
    SetupArgs sa{};
    sa.use_matrix = m ? 1 : 0;
    if (m) for (int i = 0; i < 12; i++) sa.m[i] = m[i];
    sa.verts = d_verts; sa.polys = d_polys; sa.polyverts = d_polyverts;
    sa.np = np; sa.nv = nv; sa.npv = npv;
    sa.log_focus = log_focus; sa.max_edges = max_edges;
    sa.W = c->W; sa.H = c->H;
    launch_polygon_rows(c->stream, sa, d_rows);
    HIPCHK(c, hipGetLastError());
    return LT_OK;
}

int lt_shade_object_dev(lt_ctx *c, const int32_t s9[9], int32_t x, int32_t y, int32_t z, const int32_t *d_verts, int nv,
                        const int32_t *d_normals, int nn, const int32_t *d_polys, int np, int32_t *d_polyverts,
                        const int32_t *d_polynormals, int npv, int32_t *d_polyaux, const lt_light *lights, int nlights)
{
    if (!c) return LT_ERR_ARG;
    if (np < 0 || nv < 0 || nn < 0 || npv < 0 || nlights < 0) return fail(c, LT_ERR_ARG, "lt_shade_object: negative count");
    if (nlights > LT_MAX_LIGHTS) return fail(c, LT_ERR_ARG, "lt_shade_object: %d lights, at most %d", nlights, LT_MAX_LIGHTS);
    if (np == 0) return LT_OK;
    if (!s9 || !d_verts || !d_normals || !d_polys || !d_polyverts || !d_polynormals || !d_polyaux || (nlights && !lights))
        return fail(c, LT_ERR_ARG, "lt_shade_object: NULL array");
    int rc = use_device(c);
    if (rc) return rc;
    static_assert(sizeof(lt_light) == 8 * sizeof(int32_t), "lt_light is eight ints");
    ShadeArgs a{};
    for (int i = 0; i < 9; i++) a.s[i] = s9[i];
    a.x = x; a.y = y; a.z = z;
    a.verts = d_verts; a.nv = nv; a.normals = d_normals; a.nn = nn;
    a.polys = d_polys; a.np = np; a.polyverts = d_polyverts; a.polynormals = d_polynormals; a.npv = npv;
    a.polyaux = d_polyaux;
    a.nlights = nlights;
    for (int j = 0; j < nlights; j++) memcpy(a.lights[j], &lights[j], sizeof(lt_light));
    a.status = c->status;
    launch_shade(c->stream, a);
    HIPCHK(c, hipGetLastError());
    return LT_OK;
}

int lt_shade_object(lt_ctx *c, const int32_t s9[9], int32_t x, int32_t y, int32_t z, const int32_t *verts, int nv,
                    const int32_t *normals, int nn, const int32_t *polys, int np, int32_t *polyverts,
                    const int32_t *polynormals, int npv, int32_t *polyaux, const lt_light *lights, int nlights)
{
    if (!c) return LT_ERR_ARG;
    if (np == 0) return LT_OK;
    if (np < 0 || nv <= 0 || nn <= 0 || npv <= 0 || !verts || !normals || !polys || !polyverts || !polynormals || !polyaux)
        return fail(c, LT_ERR_ARG, "lt_shade_object: bad argument");
    int rc = use_device(c);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->gs));       // earlier batches' set-up has read the staging arrays
    HIPCHK(c, hipStreamSynchronize(c->own_gs2));
    if ((rc = ensure(c, c->s_verts, (size_t)nv * 16))) return rc;
    if ((rc = ensure(c, c->s_normals, (size_t)nn * 16))) return rc;
    if ((rc = ensure(c, c->s_polys, (size_t)np * 16))) return rc;
    if ((rc = ensure(c, c->s_aux, (size_t)np * 16))) return rc;
    if ((rc = ensure(c, c->s_pv, (size_t)npv * 24))) return rc;
    if ((rc = ensure(c, c->s_pn, (size_t)npv * 4))) return rc;
    if ((rc = h2d(c, c->s_verts.p, verts, (size_t)nv * 16))) return rc;
    if ((rc = h2d(c, c->s_normals.p, normals, (size_t)nn * 16))) return rc;
    if ((rc = h2d(c, c->s_polys.p, polys, (size_t)np * 16))) return rc;
    if ((rc = h2d(c, c->s_aux.p, polyaux, (size_t)np * 16))) return rc;
    if ((rc = h2d(c, c->s_pv.p, polyverts, (size_t)npv * 24))) return rc;
    if ((rc = h2d(c, c->s_pn.p, polynormals, (size_t)npv * 4))) return rc;
    if ((rc = lt_shade_object_dev(c, s9, x, y, z, (const int *)c->s_verts.p, nv, (const int *)c->s_normals.p, nn,
                                  (const int *)c->s_polys.p, np, (int *)c->s_pv.p, (const int *)c->s_pn.p, npv,
                                  (int *)c->s_aux.p, lights, nlights)))
        return rc;
    if ((rc = d2h(c, polyverts, c->s_pv.p, (size_t)npv * 24))) return rc;
    if ((rc = d2h(c, polyaux, c->s_aux.p, (size_t)np * 16))) return rc;
    return LT_OK;
}

int lt_shade_points_dev(lt_ctx *c, const int32_t s9[9], const int32_t *d_points, const int32_t *d_normals, int n,
                        const lt_light *lights, int nlights, int32_t *d_rgb)
{
    if (!c) return LT_ERR_ARG;
    if (n < 0 || nlights < 0) return fail(c, LT_ERR_ARG, "lt_shade_points: negative count");
    if (nlights > LT_MAX_LIGHTS) return fail(c, LT_ERR_ARG, "lt_shade_points: %d lights, at most %d", nlights, LT_MAX_LIGHTS);
    if (n == 0) return LT_OK;
    if (!s9 || !d_points || !d_normals || !d_rgb || (nlights && !lights)) return fail(c, LT_ERR_ARG, "lt_shade_points: NULL array");
    int rc = use_device(c);
    if (rc) return rc;
    ShadeArgs a{};
    for (int i = 0; i < 9; i++) a.s[i] = s9[i];
    a.nlights = nlights;
    for (int j = 0; j < nlights; j++) memcpy(a.lights[j], &lights[j], sizeof(lt_light));
    a.status = c->status;
    launch_shade_points(c->stream, a, d_points, d_normals, n, d_rgb);
    HIPCHK(c, hipGetLastError());
    return LT_OK;
}

int lt_shade_points(lt_ctx *c, const int32_t s9[9], const int32_t *points, const int32_t *normals, int n,
                    const lt_light *lights, int nlights, int32_t *rgb)
{
    if (!c) return LT_ERR_ARG;
    if (n == 0) return LT_OK;
    if (n < 0 || !points || !normals || !rgb) return fail(c, LT_ERR_ARG, "lt_shade_points: bad argument");
    int rc = lt_flush(c);
    if (rc) return rc;
    if ((rc = ensure(c, c->s_verts, (size_t)n * 16))) return rc;
    if ((rc = ensure(c, c->s_normals, (size_t)n * 16))) return rc;
    if ((rc = ensure(c, c->s_aux, (size_t)n * 16))) return rc;
    if ((rc = h2d(c, c->s_verts.p, points, (size_t)n * 16))) return rc;
    if ((rc = h2d(c, c->s_normals.p, normals, (size_t)n * 16))) return rc;
    if ((rc = lt_shade_points_dev(c, s9, (const int32_t *)c->s_verts.p, (const int32_t *)c->s_normals.p, n, lights, nlights,
                                  (int32_t *)c->s_aux.p)))
        return rc;
    if ((rc = d2h(c, rgb, c->s_aux.p, (size_t)n * 16))) return rc;
    return lt_flush(c);                            // reports a point light at a point
}

int lt_plane_normals_dev(lt_ctx *c, const int32_t *d_verts, int nv, const int32_t *d_polys, int np,
                         const int32_t *d_polyverts, int npv, int32_t *d_plane_normals)
{
    if (!c) return LT_ERR_ARG;
    if (np < 0 || nv < 0 || npv < 0) return fail(c, LT_ERR_ARG, "lt_plane_normals: negative count");
    if (np == 0) return LT_OK;
    if (!d_verts || !d_polys || !d_polyverts || !d_plane_normals) return fail(c, LT_ERR_ARG, "lt_plane_normals: NULL array");
    int rc = use_device(c);
    if (rc) return rc;
    launch_plane_normals(c->stream, d_verts, nv, d_polys, np, d_polyverts, npv, d_plane_normals, c->status);
    HIPCHK(c, hipGetLastError());
    return LT_OK;
}

int lt_plane_normals(lt_ctx *c, const int32_t *verts, int nv, const int32_t *polys, int np,
                     const int32_t *polyverts, int npv, int32_t *plane_normals)
{
    if (!c) return LT_ERR_ARG;
    if (np == 0) return LT_OK;
    if (np < 0 || nv <= 0 || npv <= 0 || !verts || !polys || !polyverts || !plane_normals)
        return fail(c, LT_ERR_ARG, "lt_plane_normals: bad argument");
    int rc = lt_flush(c);                          // the staging arrays are free, earlier reports are out
    if (rc) return rc;
    if ((rc = ensure(c, c->s_verts, (size_t)nv * 16))) return rc;
    if ((rc = ensure(c, c->s_polys, (size_t)np * 16))) return rc;
    if ((rc = ensure(c, c->s_pv, (size_t)npv * 24))) return rc;
    if ((rc = ensure(c, c->s_normals, (size_t)np * 16))) return rc;
    if ((rc = h2d(c, c->s_verts.p, verts, (size_t)nv * 16))) return rc;
    if ((rc = h2d(c, c->s_polys.p, polys, (size_t)np * 16))) return rc;
    if ((rc = h2d(c, c->s_pv.p, polyverts, (size_t)npv * 24))) return rc;
    if ((rc = lt_plane_normals_dev(c, (const int *)c->s_verts.p, nv, (const int *)c->s_polys.p, np, (const int *)c->s_pv.p, npv,
                                   (int *)c->s_normals.p)))
        return rc;
    if ((rc = d2h(c, plane_normals, c->s_normals.p, (size_t)np * 16))) return rc;
    return lt_flush(c);                            // reports a zero-length normal
}

int lt_submit_object_dev(lt_ctx *c, const int32_t m[12], const int32_t *d_verts, int nv, const int32_t *d_polys,
                         const int32_t *d_polyaux, int np, const int32_t *d_polyverts, int npv, int render_type,
                         int log_focus, int force_linear, int max_edges)
{
    if (!c) return LT_ERR_ARG;
    int dim = 0, rc;
    if (render_type != LT_WIRE && (rc = check_type(c, render_type, &dim))) return rc;
    if (np < 0 || nv < 0 || npv < 0) return fail(c, LT_ERR_ARG, "lt_submit_object: negative count");
    if (np == 0) return LT_OK;
    if (!d_verts || !d_polys || !d_polyverts) return fail(c, LT_ERR_ARG, "lt_submit_object: NULL array");
    if (render_type == LT_FLAT && !d_polyaux) return fail(c, LT_ERR_ARG, "lt_submit_object: LT_FLAT needs polyaux (m_red, m_green, m_blue)");
    if (log_focus < 0 || log_focus > 30) return fail(c, LT_ERR_ARG, "lt_submit_object: log_focus %d outside 0..30", log_focus);
    if (!c->frame_defined) return fail(c, LT_ERR_ARG, "draw before the first lt_clear: the frame is undefined");
    if ((rc = use_device(c))) return rc;
    if (render_type == LT_WIRE) return run_wire(c, m, d_verts, nv, d_polys, d_polyaux, np, d_polyverts, npv, log_focus, max_edges);
    return run_batch(c, false, dim, m, d_verts, nv, d_polys, d_polyaux, np, d_polyverts, npv, render_type,
                     log_focus, force_linear, max_edges);
}

int lt_submit_objects_dev(lt_ctx *c, const lt_object *d_objects, int nobj, const int32_t *d_polyobj, const int32_t *d_verts, int nv,
                          const int32_t *d_polys, const int32_t *d_polyaux, int np, const int32_t *d_polyverts, int npv,
                          int render_type, int log_focus, int force_linear, int max_edges)
{
    if (!c) return LT_ERR_ARG;
    static_assert(sizeof(lt_object) == sizeof(ObjectDesc) && sizeof(lt_object) == 64, "lt_object is sixteen ints");
    int dim = 0, rc;
    if (render_type == LT_WIRE) return fail(c, LT_ERR_ARG, "lt_submit_objects: LT_WIRE objects go through lt_submit_object one by one");
    if ((rc = check_type(c, render_type, &dim))) return rc;
    if (np < 0 || nv < 0 || npv < 0 || nobj < 0) return fail(c, LT_ERR_ARG, "lt_submit_objects: negative count");
    if (np == 0) return LT_OK;
    if (nobj == 0 || !d_objects || !d_polyobj || !d_verts || !d_polys || !d_polyverts) return fail(c, LT_ERR_ARG, "lt_submit_objects: NULL array or empty table");
    if (((size_t)d_objects & 15) != 0) return fail(c, LT_ERR_ARG, "lt_submit_objects: the table must be 16-byte aligned");
    if (render_type == LT_FLAT && !d_polyaux) return fail(c, LT_ERR_ARG, "lt_submit_objects: LT_FLAT needs polyaux (m_red, m_green, m_blue)");
    if (log_focus < 0 || log_focus > 30) return fail(c, LT_ERR_ARG, "lt_submit_objects: log_focus %d outside 0..30", log_focus);
    if (!c->frame_defined) return fail(c, LT_ERR_ARG, "draw before the first lt_clear: the frame is undefined");
    if ((rc = use_device(c))) return rc;
    return run_batch(c, false, dim, nullptr, d_verts, nv, d_polys, d_polyaux, np, d_polyverts, npv, render_type, log_focus,
                     force_linear, max_edges, nullptr, (const ObjectDesc *)d_objects, d_polyobj, nobj);
}

int lt_submit_objects(lt_ctx *c, const lt_object *objects, int nobj, const int32_t *polyobj, const int32_t *verts, int nv,
                      const int32_t *polys, const int32_t *polyaux, int np, const int32_t *polyverts, int npv,
                      int render_type, int log_focus, int force_linear, int max_edges)
{
    if (!c) return LT_ERR_ARG;
    if (np == 0) return LT_OK;
    if (np < 0 || nv <= 0 || npv <= 0 || nobj <= 0 || !objects || !polyobj || !verts || !polys || !polyverts)
        return fail(c, LT_ERR_ARG, "lt_submit_objects: bad argument");
    int rc = use_device(c);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->gs));       // earlier batches' set-up has read the staging arrays
    HIPCHK(c, hipStreamSynchronize(c->own_gs2));
    if ((rc = ensure(c, c->s_objects, (size_t)nobj * sizeof(lt_object)))) return rc;
    if ((rc = ensure(c, c->s_polyobj, (size_t)np * 4))) return rc;
    if ((rc = ensure(c, c->s_verts, (size_t)nv * 16))) return rc;
    if ((rc = ensure(c, c->s_polys, (size_t)np * 16))) return rc;
    if ((rc = ensure(c, c->s_pv, (size_t)npv * 24))) return rc;
    if (polyaux && (rc = ensure(c, c->s_aux, (size_t)np * 16))) return rc;
    if ((rc = h2d(c, c->s_objects.p, objects, (size_t)nobj * sizeof(lt_object)))) return rc;
    if ((rc = h2d(c, c->s_polyobj.p, polyobj, (size_t)np * 4))) return rc;
    if ((rc = h2d(c, c->s_verts.p, verts, (size_t)nv * 16))) return rc;
    if ((rc = h2d(c, c->s_polys.p, polys, (size_t)np * 16))) return rc;
    if ((rc = h2d(c, c->s_pv.p, polyverts, (size_t)npv * 24))) return rc;
    if (polyaux && (rc = h2d(c, c->s_aux.p, polyaux, (size_t)np * 16))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return lt_submit_objects_dev(c, (const lt_object *)c->s_objects.p, nobj, (const int32_t *)c->s_polyobj.p, (const int *)c->s_verts.p, nv,
                                 (const int *)c->s_polys.p, polyaux ? (const int *)c->s_aux.p : nullptr, np, (const int *)c->s_pv.p,
                                 npv, render_type, log_focus, force_linear, max_edges);
}

int lt_submit_object_dev_noalloc(lt_ctx *c, const int32_t m[12], const int32_t *d_verts, int nv, const int32_t *d_polys,
                                 const int32_t *d_polyaux, int np, const int32_t *d_polyverts, int npv, int render_type,
                                 int log_focus, int force_linear, int max_edges)
{
    if (!c) return LT_ERR_ARG;
    int dim = 0, rc;
    if (render_type == LT_WIRE) return fail(c, LT_ERR_ARG, "lt_submit_object_dev_noalloc: LT_WIRE is not supported (use lt_submit_object_dev)");
    if ((rc = check_type(c, render_type, &dim))) return rc;
    if (np < 0 || nv < 0 || npv < 0) return fail(c, LT_ERR_ARG, "lt_submit_object_dev_noalloc: negative count");
    if (np == 0) return LT_OK;
    if (!d_verts || !d_polys || !d_polyverts) return fail(c, LT_ERR_ARG, "lt_submit_object_dev_noalloc: NULL array");
    if (render_type == LT_FLAT && !d_polyaux) return fail(c, LT_ERR_ARG, "lt_submit_object_dev_noalloc: LT_FLAT needs polyaux (m_red, m_green, m_blue)");
    if (log_focus < 0 || log_focus > 30) return fail(c, LT_ERR_ARG, "lt_submit_object_dev_noalloc: log_focus %d outside 0..30", log_focus);
    if (!c->frame_defined) return fail(c, LT_ERR_ARG, "draw before the first lt_clear: the frame is undefined");
    if ((rc = use_device(c))) return rc;
    NoAllocDevice &dev = g_noalloc[c->device];
    if (!dev.ptr) return fail(c, LT_ERR_HIP, "lt_submit_object_dev_noalloc: the static work array has no address on device %d", c->device);

    // carve the work arrays out of the static array, 256 bytes apart at least
    // tile-list capacity: the same minimum as lt_submit_object_dev (2^20 entries, 4 MB), so that a
    // batch of few but large polygons draws the same; only where that does not fit the array
    // beside the batch's other arrays, the smaller minimum
    BatchPlan pl = plan_batch(c, false, dim, np, max_edges, 1u << 20);
    {
        const BatchPlan small = plan_batch(c, false, dim, np, max_edges, 1u << 16);
        auto total = [](const BatchPlan &q) {
            const size_t parts[] = {q.flags, q.tcount, q.toffset, q.tcursor, q.hdr, q.edges, q.list, q.want_prp ? q.prp : 0, q.queue, q.sq_queue, q.sq_desc, q.sq_tuples};
            size_t t = 0;
            for (size_t b : parts) t += (b + 255) & ~(size_t)255;
            return t;
        };
        if (total(pl) > (size_t)LT_NOALLOC_SCRATCH_BYTES && total(small) <= (size_t)LT_NOALLOC_SCRATCH_BYTES) pl = small;
    }
    WorkArrays wa;
    size_t used = 0;
    unsigned char *base = (unsigned char *)dev.ptr;
    auto take = [&](size_t bytes) { void *p = base + used; used += (bytes + 255) & ~(size_t)255; return p; };
    // the arrays that must start at zero come first, so one fill covers both
    wa.flags = (int *)take(pl.flags);
    wa.tcount = (unsigned *)take(pl.tcount);
    const size_t zeroed = used;
    wa.toffset = (unsigned *)take(pl.toffset);
    wa.tcursor = (unsigned *)take(pl.tcursor);
    wa.hdr = (PrimHdr *)take(pl.hdr);
    wa.edges = (int *)take(pl.edges);
    wa.list = (unsigned *)take(pl.list);
    if (pl.want_prp) wa.prp = (int *)take(pl.prp);
    wa.queue = (uint2 *)take(pl.queue);
    wa.sq_queue = (uint2 *)take(pl.sq_queue);
    wa.sq_desc = (int *)take(pl.sq_desc);
    wa.sq_tuples = (int *)take(pl.sq_tuples);
    if (used > (size_t)LT_NOALLOC_SCRATCH_BYTES)
        return fail(c, LT_ERR_CAPACITY,
                    "lt_submit_object_dev_noalloc: %d polygons (max_edges %d, %d x %d screen, %llu list entries) need %zu bytes of "
                    "work arrays, the static array holds %zu: submit fewer polygons per call, lower max_edges or "
                    "lt_set_list_capacity, or use lt_submit_object_dev",
                    np, pl.max_edges, c->W, c->H, (unsigned long long)pl.capacity, used, (size_t)LT_NOALLOC_SCRATCH_BYTES);

    // Every kernel of the batch goes on the stream of the current frame's fills: after the
    // fills submitted before it, before those submitted after it.
    hipStream_t s = c->rs;
    if (c->pipelined && hipStreamQuery(c->stream) != hipSuccess) {
        (void)hipGetLastError();
        HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));       // the inputs may come from the caller's stream
        HIPCHK(c, hipStreamWaitEvent(s, c->ev_fork, 0));
    }
    // the array is shared by every call on this GPU: wait for the batch that used it last
    // (it may be on the other frame's stream, or another context's); from this wait to the
    // record below no other host thread may enqueue a batch that uses the array
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (dev.pending) HIPCHK(c, hipStreamWaitEvent(s, dev.done, 0));
    HIPCHK(c, hipMemsetAsync(base, 0, zeroed, s));              // order flag, sub-queue counters, tile counts
    rc = issue_batch(c, wa, pl, s, s, nullptr, nullptr, false, false, dim, m, d_verts, nv, d_polys, d_polyaux, np, d_polyverts,
                     npv, render_type, log_focus, force_linear);
    // recorded even after a failed launch: what was issued before the failure still uses the array
    hipError_t e = hipEventRecord(dev.done, s);
    dev.pending = true;
    if (rc) return rc;
    HIPCHK(c, e);
    return LT_OK;
}

int lt_submit_object(lt_ctx *c, const int32_t m[12], const int32_t *verts, int nv, const int32_t *polys,
                     const int32_t *polyaux, int np, const int32_t *polyverts, int npv, int render_type,
                     int log_focus, int force_linear, int max_edges)
{
    if (!c) return LT_ERR_ARG;
    if (np == 0) return LT_OK;
    if (np < 0 || nv <= 0 || npv <= 0 || !verts || !polys || !polyverts)
        return fail(c, LT_ERR_ARG, "lt_submit_object: bad argument");
    int rc = use_device(c);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->gs));       // earlier batches' set-up has read the staging arrays
    HIPCHK(c, hipStreamSynchronize(c->own_gs2));
    if ((rc = ensure(c, c->s_verts, (size_t)nv * 16))) return rc;
    if ((rc = ensure(c, c->s_polys, (size_t)np * 16))) return rc;
    if ((rc = ensure(c, c->s_pv, (size_t)npv * 24))) return rc;
    if (polyaux && (rc = ensure(c, c->s_aux, (size_t)np * 16))) return rc;
    // on return from h2d the source has been read (or is pinned memory, read before the wait below ends)
    if ((rc = h2d(c, c->s_verts.p, verts, (size_t)nv * 16))) return rc;
    if ((rc = h2d(c, c->s_polys.p, polys, (size_t)np * 16))) return rc;
    if ((rc = h2d(c, c->s_pv.p, polyverts, (size_t)npv * 24))) return rc;
    if (polyaux && (rc = h2d(c, c->s_aux.p, polyaux, (size_t)np * 16))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    rc = lt_submit_object_dev(c, m, (const int *)c->s_verts.p, nv, (const int *)c->s_polys.p,
                              polyaux ? (const int *)c->s_aux.p : nullptr, np, (const int *)c->s_pv.p, npv,
                              render_type, log_focus, force_linear, max_edges);
    // a wireframe batch reads the staging arrays on the stream of the fills
    if (rc == LT_OK && render_type == LT_WIRE) HIPCHK(c, hipStreamSynchronize(c->rs));
    return rc;
}

int lt_submit_screen_polys(lt_ctx *c, int kind, const int32_t *tuples, int ntuples, const int32_t *desc, int n,
                           int max_edges)
{
    if (!c) return LT_ERR_ARG;
    int dim = 0, rc;
    if ((rc = check_type(c, kind, &dim))) return rc;
    if (n == 0) return LT_OK;
    if (n < 0 || ntuples <= 0 || !tuples || !desc) return fail(c, LT_ERR_ARG, "lt_submit_screen_polys: bad argument");
    if (!c->frame_defined) return fail(c, LT_ERR_ARG, "draw before the first lt_clear: the frame is undefined");
    if ((rc = use_device(c))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->gs));       // earlier batches' set-up has read the staging arrays
    HIPCHK(c, hipStreamSynchronize(c->own_gs2));
    if ((rc = ensure(c, c->s_polys, (size_t)n * 16))) return rc;
    if ((rc = ensure(c, c->s_pv, (size_t)ntuples * dim * 4))) return rc;
    if ((rc = h2d(c, c->s_polys.p, desc, (size_t)n * 16))) return rc;
    if ((rc = h2d(c, c->s_pv.p, tuples, (size_t)ntuples * dim * 4))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return run_batch(c, true, dim, nullptr, nullptr, 0, (const int *)c->s_polys.p, nullptr, n, (const int *)c->s_pv.p, ntuples,
                     kind, 0, 0, max_edges);
}

int lt_submit_screen_polys_prp(lt_ctx *c, const int32_t *tuples, int ntuples, const int32_t *desc, const int32_t *prp, int n,
                               int log_focus, int max_edges)
{
    if (!c) return LT_ERR_ARG;
    if (n == 0) return LT_OK;
    if (n < 0 || ntuples <= 0 || !tuples || !desc || !prp) return fail(c, LT_ERR_ARG, "lt_submit_screen_polys_prp: bad argument");
    if (log_focus < 0 || log_focus > 30) return fail(c, LT_ERR_ARG, "lt_submit_screen_polys_prp: log_focus %d outside 0..30", log_focus);
    if (!c->frame_defined) return fail(c, LT_ERR_ARG, "draw before the first lt_clear: the frame is undefined");
    int rc;
    if ((rc = use_device(c))) return rc;
    // descriptors with the perspective fill named in the length word; the nine coefficients of
    // GraphicsPoly.cpp:570-578 from O, u, v in wrapping 32-bit arithmetic, as the set-up kernel
    // makes them for object batches
    std::vector<int32_t> d((size_t)n * 4), k((size_t)n * PRP_INTS);
    auto wm = [](int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); };
    auto ws = [](int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); };
    for (int i = 0; i < n; i++) {
        const int32_t *q = prp + (size_t)i * 12;
        const int32_t len = desc[(size_t)i * 4 + 1];
        if (len < 0 || len > 255) return fail(c, LT_ERR_ARG, "lt_submit_screen_polys_prp: polygon %d has %d edges", i, len);
        if (q[11] < 0 || q[11] > 24) return fail(c, LT_ERR_ARG, "lt_submit_screen_polys_prp: polygon %d: log_texture_scale %d outside 0..24", i, q[11]);
        d[(size_t)i * 4] = desc[(size_t)i * 4];
        d[(size_t)i * 4 + 1] = len | (9 << 8);
        d[(size_t)i * 4 + 2] = desc[(size_t)i * 4 + 2];
        d[(size_t)i * 4 + 3] = desc[(size_t)i * 4 + 3];
        const int32_t x0 = q[0], y0 = q[1], z0 = q[2], u0 = q[3], v0 = q[4];
        const int32_t *u = q + 5, *v = q + 8;
        int32_t *o = k.data() + (size_t)i * PRP_INTS;
        o[0] = ws(wm(v[2], y0), wm(v[1], z0));      // ui
        o[1] = ws(wm(v[0], z0), wm(v[2], x0));      // uj
        o[2] = ws(wm(v[1], x0), wm(v[0], y0));      // uc
        o[3] = ws(wm(u[1], z0), wm(u[2], y0));      // vi
        o[4] = ws(wm(u[2], x0), wm(u[0], z0));      // vj
        o[5] = ws(wm(u[0], y0), wm(u[1], x0));      // vc
        o[6] = ws(wm(v[1], u[2]), wm(v[2], u[1]));  // zi
        o[7] = ws(wm(v[2], u[0]), wm(v[0], u[2]));  // zj
        o[8] = ws(wm(v[0], u[1]), wm(v[1], u[0]));  // zc
        o[9] = u0; o[10] = v0; o[11] = q[11];
    }
    HIPCHK(c, hipStreamSynchronize(c->gs));       // earlier batches' set-up has read the staging arrays
    HIPCHK(c, hipStreamSynchronize(c->own_gs2));
    if ((rc = ensure(c, c->s_polys, (size_t)n * 16))) return rc;
    if ((rc = ensure(c, c->s_pv, (size_t)ntuples * 8 * 4))) return rc;
    if ((rc = ensure(c, c->s_prp, (size_t)n * PRP_INTS * 4))) return rc;
    if ((rc = h2d(c, c->s_polys.p, d.data(), (size_t)n * 16))) return rc;
    if ((rc = h2d(c, c->s_pv.p, tuples, (size_t)ntuples * 8 * 4))) return rc;
    if ((rc = h2d(c, c->s_prp.p, k.data(), (size_t)n * PRP_INTS * 4))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return run_batch(c, true, 8, nullptr, nullptr, 0, (const int *)c->s_polys.p, nullptr, n, (const int *)c->s_pv.p, ntuples,
                     LT_TEXTURED, log_focus, 0, max_edges, (const int *)c->s_prp.p);
}

int lt_submit_lines_dev(lt_ctx *c, const int32_t *d_lines, int n)
{
    if (!c) return LT_ERR_ARG;
    if (n == 0) return LT_OK;
    if (n < 0 || !d_lines) return fail(c, LT_ERR_ARG, "lt_submit_lines: bad argument");
    if (((size_t)d_lines & 15) != 0) return fail(c, LT_ERR_ARG, "lt_submit_lines: the array must be 16-byte aligned");
    if (!c->frame_defined) return fail(c, LT_ERR_ARG, "draw before the first lt_clear: the frame is undefined");
    int rc = use_device(c);
    if (rc) return rc;
    hipStream_t s;
    if ((rc = wire_begin(c, &s))) return rc;
    launch_lines(s, d_lines, n, c->W, c->H, c->band_y0, c->band_y1, c->clip, c->zbuf, (unsigned *)c->line_winner.p, c->cbuf, c->status);
    HIPCHK(c, hipGetLastError());
    return LT_OK;
}

int lt_submit_lines(lt_ctx *c, const int32_t *lines, int n)
{
    if (!c) return LT_ERR_ARG;
    if (n == 0) return LT_OK;
    if (n < 0 || !lines) return fail(c, LT_ERR_ARG, "lt_submit_lines: bad argument");
    int rc = use_device(c);
    if (rc) return rc;
    // an earlier list may still be read, by the line kernels of this frame or of the frame
    // before it, which run on the other fill stream when two frames are in flight
    HIPCHK(c, hipStreamSynchronize(c->rs));
    if (c->own_rs && c->own_rs != c->rs) HIPCHK(c, hipStreamSynchronize(c->own_rs));
    if (c->alt_rs && c->alt_rs != c->rs) HIPCHK(c, hipStreamSynchronize(c->alt_rs));
    if ((rc = ensure(c, c->s_lines, (size_t)n * 32))) return rc;
    if ((rc = h2d(c, c->s_lines.p, lines, (size_t)n * 32))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return lt_submit_lines_dev(c, (const int32_t *)c->s_lines.p, n);
}

}  // extern "C"
