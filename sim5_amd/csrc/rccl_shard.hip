// rccl_shard.hip -- multi-GPU thin-disk image job over RCCL (include/sim5gpu_rccl.h): host code only, on top of the
// C-ABI of libsim5gpu.so (the tracing and placement kernels) and librccl.
//
// Per image and rank: ONE tracing launch of the rank's mirrored stripe pairs, ONE ncclGather (in place on the root: its
// own block of the receive buffer is its send buffer, so nothing of the root's is copied -- its rows are traced straight
// into the image), and on the root one launch for its band and one placement launch.  The gather runs on a
// communication stream of its own and is tied to the caller's stream by events, so tracing image i+1 overlaps it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include "../../include/sim5gpu_rccl.h"

static thread_local char g_rccl_err[512] = "";

static int fail_hip(const char* what, hipError_t e)
{
    snprintf(g_rccl_err, sizeof g_rccl_err, "%s: %s", what, hipGetErrorString(e));
    return SIM5GPU_E_HIP;
}
static int fail_nccl(const char* what, ncclResult_t r)
{
    snprintf(g_rccl_err, sizeof g_rccl_err, "%s: %s", what, ncclGetErrorString(r));
    return SIM5GPU_E_RCCL;
}
static int fail_base(const char* what, int rc)
{
    snprintf(g_rccl_err, sizeof g_rccl_err, "%s failed (%d): %s", what, rc, sim5gpu_last_error());
    return rc;
}
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail_hip(#call, e_); } while (0)
#define NCCLCHK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return fail_nccl(#call, r_); } while (0)

static int upper_half(int ny) { return (ny + 1) / 2; }

struct sim5gpu_shard {
    ncclComm_t comm;
    int rank, world, nx, ny, dealt;
    int rows_max;                         // rows of the largest share: every rank sends 2 x rows_max x nx floats
    size_t block;                         // floats per rank and image
    float* payload[2];                    // peers: the share, packed; rank 0: [world] blocks, block 0 unused
    hipStream_t comm_stream;
    hipEvent_t traced[2], gathered[2];
    float* image_f[2];                    // rank 0: planes of the image each slot in flight belongs to
    float* image_g[2];
    sim5gpu_image_desc peers[16];         // rank 0: job descriptions (row geometry) of ranks 1 .. world-1
    int n_peers;
    unsigned long long begun, ended;
    int poisoned;                         // a launch of this rank failed after validation: its rows of some image are not valid
};

extern "C" {

const char* sim5gpu_rccl_last_error(void) { return g_rccl_err; }

int sim5gpu_rccl_unique_id(void* id128)
{
    if (!id128) return SIM5GPU_E_ARG;
    ncclUniqueId id;
    NCCLCHK(ncclGetUniqueId(&id));
    static_assert(sizeof id == SIM5GPU_RCCL_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, &id, sizeof id);
    return SIM5GPU_OK;
}

int sim5gpu_rccl_comm_create(const void* id128, int rank, int world, void** comm)
{
    if (!id128 || !comm || world < 1 || rank < 0 || rank >= world) return SIM5GPU_E_ARG;
    if (sim5gpu_device_count() < 1) { snprintf(g_rccl_err, sizeof g_rccl_err, "no HIP device visible; there is no CPU fallback"); return SIM5GPU_E_NO_DEVICE; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    NCCLCHK(ncclCommInitRank(&c, world, id, rank));
    *comm = (void*)c;
    return SIM5GPU_OK;
}

int sim5gpu_rccl_comm_destroy(void* comm)
{
    if (!comm) return SIM5GPU_OK;
    NCCLCHK(ncclCommDestroy((ncclComm_t)comm));
    return SIM5GPU_OK;
}

int sim5gpu_shard_plan(const sim5gpu_image_desc* image, int rank, int world, int dealt_rows,
                       int* rows_traced, int* band_y0, int* band_y1, sim5gpu_image_desc* share)
{
    if (!image || world < 1 || world > 16 || rank < 0 || rank >= world || image->nx <= 0 || image->ny <= 0) {
        snprintf(g_rccl_err, sizeof g_rccl_err, "shard_plan: bad arguments (1 <= world <= 16, 0 <= rank < world, nx, ny > 0)");
        return SIM5GPU_E_ARG;
    }
    const int ny = image->ny, half = upper_half(ny);
    const int dealt = (dealt_rows <= 0 || dealt_rows >= half || world == 1) ? half : dealt_rows;
    sim5gpu_image_desc d = *image;
    d.y0 = rank * SIM5GPU_SHARD_STRIPE_ROWS;
    d.y1 = dealt;
    d.stripe_rows = SIM5GPU_SHARD_STRIPE_ROWS;
    d.stripe_step = world * SIM5GPU_SHARD_STRIPE_ROWS;
    if (image->flags & ~(SIM5GPU_IMG_STRICT | SIM5GPU_IMG_DIRECT)) {
        snprintf(g_rccl_err, sizeof g_rccl_err, "shard_plan: the whole-image description may carry SIM5GPU_IMG_STRICT and SIM5GPU_IMG_DIRECT only (flags 0x%x)", image->flags);
        return SIM5GPU_E_ARG;
    }
    // the arithmetic variant of the caller's description travels to every launch of the split (share and band)
    d.flags = (image->flags & (SIM5GPU_IMG_STRICT | SIM5GPU_IMG_DIRECT)) | SIM5GPU_IMG_MIRROR | (rank == 0 ? SIM5GPU_IMG_INPLACE : 0);
    int rows = (d.y0 < d.y1) ? sim5gpu_image_rows(&d) : 0;
    int b0 = 0, b1 = 0;
    if (dealt < half && ny - dealt > dealt) { b0 = dealt; b1 = ny - dealt; }
    if (rank == 0) rows += b1 - b0;
    if (rows_traced) *rows_traced = rows;
    if (band_y0) *band_y0 = b0;
    if (band_y1) *band_y1 = b1;
    if (share) *share = d;
    return SIM5GPU_OK;
}

int sim5gpu_shard_create(sim5gpu_shard** shard, void* comm, int rank, int world, int nx, int ny, int dealt_rows)
{
    if (!shard || (!comm && world > 1) || world < 1 || world > 16 || rank < 0 || rank >= world || nx <= 0 || ny <= 0) {
        snprintf(g_rccl_err, sizeof g_rccl_err, "shard_create: bad arguments");
        return SIM5GPU_E_ARG;
    }
    if (sim5gpu_device_count() < 1) { snprintf(g_rccl_err, sizeof g_rccl_err, "no HIP device visible; there is no CPU fallback"); return SIM5GPU_E_NO_DEVICE; }
    sim5gpu_shard* s = new (std::nothrow) sim5gpu_shard();
    if (!s) return SIM5GPU_E_ARG;
    memset(s, 0, sizeof *s);
    s->comm = (ncclComm_t)comm; s->rank = rank; s->world = world; s->nx = nx; s->ny = ny;
    const int half = upper_half(ny);
    s->dealt = (dealt_rows <= 0 || dealt_rows >= half || world == 1) ? half : dealt_rows;
    sim5gpu_image_desc whole;
    memset(&whole, 0, sizeof whole);
    whole.nx = nx; whole.ny = ny; whole.y0 = 0; whole.y1 = ny;
    s->rows_max = 1;
    for (int r = 0; r < world; ++r) {
        sim5gpu_image_desc d;
        int rows = 0, b0, b1;
        int rc = sim5gpu_shard_plan(&whole, r, world, s->dealt, &rows, &b0, &b1, &d);
        if (rc) { delete s; return rc; }
        const int own = (d.y0 < d.y1) ? sim5gpu_image_rows(&d) : 0;
        if (own > s->rows_max) s->rows_max = own;
        if (r > 0) {
            if (own == 0) {                                     // the placement kernel wants every share non-empty
                snprintf(g_rccl_err, sizeof g_rccl_err, "shard_create: rank %d of %d has no rows of a %d-row image (fewer stripes than ranks)", r, world, ny);
                delete s;
                return SIM5GPU_E_ARG;
            }
            d.flags &= ~SIM5GPU_IMG_INPLACE;
            s->peers[s->n_peers++] = d;
        }
    }
    s->block = (size_t)2 * (size_t)s->rows_max * (size_t)nx;
    hipError_t e = hipSuccess;
    const size_t bytes = s->block * sizeof(float) * (rank == 0 ? (size_t)world : 1);
    for (int b = 0; b < 2 && e == hipSuccess; ++b) {
        e = hipMalloc((void**)&s->payload[b], bytes);
        if (e == hipSuccess) e = hipMemset(s->payload[b], 0, bytes);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->traced[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->gathered[b], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { sim5gpu_shard_destroy(s); return fail_hip("shard_create", e); }
    *shard = s;
    return SIM5GPU_OK;
}

int sim5gpu_shard_destroy(sim5gpu_shard* s)
{
    if (!s) return SIM5GPU_OK;
    (void)hipDeviceSynchronize();
    for (int b = 0; b < 2; ++b) {
        if (s->payload[b]) (void)hipFree(s->payload[b]);
        if (s->traced[b]) (void)hipEventDestroy(s->traced[b]);
        if (s->gathered[b]) (void)hipEventDestroy(s->gathered[b]);
    }
    if (s->comm_stream) (void)hipStreamDestroy(s->comm_stream);
    delete s;
    return SIM5GPU_OK;
}

int sim5gpu_shard_image_begin(sim5gpu_shard* s, const sim5gpu_image_desc* image, float* d_image_f, float* d_image_g, void* stream)
{
    if (!s || !image) return SIM5GPU_E_ARG;
    if (image->nx != s->nx || image->ny != s->ny || image->y0 != 0 || image->y1 != image->ny || image->stripe_rows != 0 ||
        (image->flags & (SIM5GPU_IMG_MIRROR | SIM5GPU_IMG_INPLACE))) {
        snprintf(g_rccl_err, sizeof g_rccl_err, "shard_image: pass the WHOLE-image description (%d x %d, all rows, no striping)", s->nx, s->ny);
        return SIM5GPU_E_ARG;
    }
    if (s->rank == 0 && (!d_image_f || !d_image_g)) { snprintf(g_rccl_err, sizeof g_rccl_err, "shard_image: rank 0 needs the image planes"); return SIM5GPU_E_ARG; }
    if (s->begun - s->ended >= 2) { snprintf(g_rccl_err, sizeof g_rccl_err, "shard_image_begin: two images are in flight already: call shard_image_end"); return SIM5GPU_E_ARG; }
    const int b = (int)(s->begun & 1ull);
    hipStream_t st = (hipStream_t)stream;
    // ---- everything that can be refused is checked BEFORE anything is enqueued: a rank that returns from here has not
    //      touched the collective, and by the same test on the same description every rank returns (or none does)
    sim5gpu_image_desc d, band;
    int rows, b0, b1, rc;
    if ((rc = sim5gpu_shard_plan(image, s->rank, s->world, s->dealt, &rows, &b0, &b1, &d)) != 0) return rc;
    const bool have_rows = d.y0 < d.y1;
    const bool have_band = (s->rank == 0 && b1 > b0);
    {   // the rank-independent part (physics, disk, image size) on the whole-image description, then this rank's own launches
        if ((rc = sim5gpu_image_desc_check(image)) != 0) return fail_base("shard_image_begin: image description", rc);
        if (have_rows && (rc = sim5gpu_image_desc_check(&d)) != 0) return fail_base("shard_image_begin: share description", rc);
        band = *image;
        band.y0 = b0; band.y1 = b1;
        if (have_band && (rc = sim5gpu_image_desc_check(&band)) != 0) return fail_base("shard_image_begin: band description", rc);
    }
    // ---- from here on the rank ALWAYS joins the gather: a launch that fails now (a HIP error) poisons the shard and is
    //      reported, but the peers are not left waiting in a collective
    int bad = 0;
    if (s->rank != 0) {
        float* pf = s->payload[b];
        if (have_rows && (rc = sim5gpu_disk_image(&d, pf, pf + (size_t)s->rows_max * (size_t)s->nx, nullptr, stream)) != 0) bad = fail_base("sim5gpu_disk_image (share)", rc);
    } else {
        s->image_f[b] = d_image_f; s->image_g[b] = d_image_g;
    }
    if (s->comm) {                                            // also with a world of one: the collective degenerates, the path is the same
        // peers: the gather after their share has been traced.  The root only RECEIVES (in place: sendbuff == recvbuff + rank *
        // sendcount, its own block is never read), so its gather does not depend on its tracing and is issued BEFORE it: the
        // event orders it after what `stream` held so far (the placement that last read this payload slot).
        hipError_t e = hipEventRecord(s->traced[b], st);
        if (e == hipSuccess) e = hipStreamWaitEvent(s->comm_stream, s->traced[b], 0);
        if (e != hipSuccess && !bad) bad = fail_hip("shard_image_begin: event before the gather", e);
        const ncclResult_t r = ncclGather(s->payload[b], s->rank == 0 ? s->payload[b] : nullptr, s->block, ncclFloat, 0, s->comm, s->comm_stream);
        if (r != ncclSuccess) { s->poisoned = 1; return fail_nccl("ncclGather", r); }      // not enqueued: the slot is not in flight
        s->begun++;                                           // the gather is in flight: the slot is tracked from here, whatever follows
        e = hipEventRecord(s->gathered[b], s->comm_stream);
        if (e != hipSuccess && !bad) bad = fail_hip("shard_image_begin: event after the gather", e);
    } else {
        s->begun++;
    }
    if (s->rank == 0 && !bad) {
        // the root's own rows while the gather is in flight: its share in place and its band, ONE job-list launch
        sim5gpu_image_desc jobs[2];
        float* jf[2]; float* jg[2];
        int nj = 0;
        if (have_rows) { jobs[nj] = d; jf[nj] = d_image_f; jg[nj] = d_image_g; ++nj; }
        if (have_band) { const size_t off = (size_t)b0 * (size_t)s->nx; jobs[nj] = band; jf[nj] = d_image_f + off; jg[nj] = d_image_g + off; ++nj; }
        if (nj > 0 && (rc = sim5gpu_disk_image_jobs(nj, jobs, jf, jg, stream)) != 0) bad = fail_base("sim5gpu_disk_image_jobs (share of rank 0 in place + band)", rc);
    }
    if (bad) { s->poisoned = 1; return bad; }                 // the image in this slot is not valid; shard_image_end still has to be called for it
    return SIM5GPU_OK;
}

int sim5gpu_shard_image_end(sim5gpu_shard* s, void* stream)
{
    if (!s) return SIM5GPU_E_ARG;
    if (s->ended >= s->begun) { snprintf(g_rccl_err, sizeof g_rccl_err, "shard_image_end: no image in flight"); return SIM5GPU_E_ARG; }
    const int b = (int)(s->ended & 1ull);
    if (s->comm) {
        HIPCHK(hipStreamWaitEvent((hipStream_t)stream, s->gathered[b], 0));
        if (s->rank == 0 && s->n_peers > 0) {
            const int rc = sim5gpu_image_place_shares(s->n_peers, s->peers, s->payload[b] + s->block, (size_t)s->rows_max,
                                                      s->image_f[b], s->image_g[b], stream);
            if (rc) return fail_base("sim5gpu_image_place_shares", rc);
        }
    }
    s->ended++;
    return SIM5GPU_OK;
}

int sim5gpu_disk_image_sharded(sim5gpu_shard* s, const sim5gpu_image_desc* image, float* d_image_f, float* d_image_g, void* stream)
{
    if (!s) return SIM5GPU_E_ARG;
    const unsigned long long before = s->begun;
    const int rc = sim5gpu_shard_image_begin(s, image, d_image_f, d_image_g, stream);
    if (rc && s->begun == before) return rc;                  // refused before anything was enqueued
    const int rc2 = sim5gpu_shard_image_end(s, stream);       // the slot is in flight: it is ended either way
    return rc ? rc : rc2;
}

/* 1 if a launch of this rank failed after its gather had been joined (some image holds invalid rows of this rank) */
int sim5gpu_shard_poisoned(const sim5gpu_shard* s) { return s ? s->poisoned : 0; }

} // extern "C"
