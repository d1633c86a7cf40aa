/*
 * sim5gpu_rccl.h -- C-ABI of the multi-GPU form of the thin-disk image job: one process per GPU, the image rows dealt
 * to the ranks as mirrored 64-row stripe pairs, ONE RCCL gather per image over xGMI, the image assembled in row order on
 * rank 0 (BASELINE.json north_star: "shards by row-tile across the 8 GPUs of one node with a single RCCL gather";
 * SURVEY.md 8(e); /opt/rocm/include/rccl/rccl.h:745 ncclGather).
 *
 * Implemented by sim5_amd/lib/libsim5gpu_rccl.so, a separate library on top of libsim5gpu.so so that the base library
 * keeps no RCCL dependency.  Plain C: pointers, sizes, PODs; the communicator crosses the boundary as a void* that
 * holds an ncclComm_t (a caller that already has one -- MPI + ncclCommInitRank -- passes its own).
 *
 * The reference has no multi-process form (README.md:202: "it does not have any parallelization built in itself"): the
 * caller loop this replaces is the single-process pixel loop of ref examples/04-disk-image-eqplane/disk-image.c:53-105,
 * now spread over the ranks.  Rays are independent, so no exchange happens while tracing.
 */
#ifndef SIM5GPU_RCCL_H
#define SIM5GPU_RCCL_H

#include "sim5gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SIM5GPU_E_RCCL       -5    /* an RCCL call failed (see sim5gpu_rccl_last_error)            */
#define SIM5GPU_RCCL_ID_BYTES 128  /* = NCCL_UNIQUE_ID_BYTES                                        */
#define SIM5GPU_SHARD_STRIPE_ROWS 64

const char *sim5gpu_rccl_last_error(void);

/* Communicator helpers for callers that have none: rank 0 makes the id, every rank receives the 128 bytes by the
 * caller's own means (a file, MPI_Bcast, a socket) and joins on ITS current device.  A process that touches the GPU must
 * not be replaced by exec: start the ranks (mpirun, torchrun, a shell loop) before any of these calls. */
int sim5gpu_rccl_unique_id(void *id128);
int sim5gpu_rccl_comm_create(const void *id128, int rank, int world, void **comm);
int sim5gpu_rccl_comm_destroy(void *comm);

/* The dealing rule, host arithmetic only (no GPU, no communicator): rows of the upper half that are dealt when a
 * band is kept on rank 0 (dealt_rows <= 0 or >= (ny+1)/2: everything is dealt), the rows rank `rank` traces per image
 * (its stripes + their mirrors, + the band on rank 0), the band [band_y0, band_y1) (empty: both 0), and the job
 * description of the rank's share derived from the whole-image description (`share` may be NULL). */
int sim5gpu_shard_plan(const sim5gpu_image_desc *image, int rank, int world, int dealt_rows,
                       int *rows_traced, int *band_y0, int *band_y1, sim5gpu_image_desc *share);

typedef struct sim5gpu_shard sim5gpu_shard;          /* buffers, streams and events of one rank */

/* Create the rank's state for images of nx x ny pixels (allocates the gather payload: 2 x rows_max x nx floats, double
 * buffered; on rank 0 world times that).  Collective in the sense that every rank must use the same nx, ny, dealt_rows. */
int sim5gpu_shard_create(sim5gpu_shard **shard, void *comm, int rank, int world, int nx, int ny, int dealt_rows);
int sim5gpu_shard_destroy(sim5gpu_shard *shard);

/* One image, every rank calls it with the same whole-image description (y0 = 0, y1 = ny, no striping; spin, inclination,
 * disk parameters may differ from image to image).  Asynchronous on `stream` (the same stream for every call of a shard):
 *   begin:  this rank's stripes (the pairing kernel), the gather on the library's communication stream once they are
 *           traced, and on rank 0 its rows in place in d_image_f / d_image_g (ny x nx planes; NULL on the other ranks) and
 *           its band while the gather is in flight;
 *   end:    `stream` waits for the gather of the OLDEST image begun, rank 0 puts the peers' rows at their image rows
 *           (sim5gpu_image_place_shares): work enqueued on `stream` after end() sees that image complete.
 * Two images may be in flight: begin(i+1) before end(i) overlaps the gather of image i with the tracing of image i+1.
 * sim5gpu_disk_image_sharded = begin + end.
 * Failure modes: begin() validates the description and this rank's launches before it enqueues anything -- every rank
 * applies the same tests to the same description, so a refused image is refused by all and no collective is entered.  A
 * HIP error after that point is reported, but the rank still joins the gather (its peers are never left waiting); the
 * shard is then marked (sim5gpu_shard_poisoned) and the image in that slot holds invalid rows of this rank: callers agree
 * on the status across ranks (MPI_Allreduce of the return codes, ...) before they use an image.  The whole-image
 * description may carry SIM5GPU_IMG_STRICT and SIM5GPU_IMG_DIRECT; both reach every launch of the split. */
int sim5gpu_shard_image_begin(sim5gpu_shard *shard, const sim5gpu_image_desc *image, float *d_image_f, float *d_image_g,
                              void *stream);
int sim5gpu_shard_image_end(sim5gpu_shard *shard, void *stream);
int sim5gpu_disk_image_sharded(sim5gpu_shard *shard, const sim5gpu_image_desc *image, float *d_image_f, float *d_image_g,
                               void *stream);
int sim5gpu_shard_poisoned(const sim5gpu_shard *shard);

#ifdef __cplusplus
}
#endif
#endif /* SIM5GPU_RCCL_H */
