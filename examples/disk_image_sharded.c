/*
 * disk_image_sharded.c -- the thin-disk image of the reference example (ref examples/04-disk-image-eqplane/disk-image.c:
 * 53-105) on N GPUs from C: one process per GPU, each tracing its mirrored row stripes, ONE RCCL gather per image, the
 * image assembled on rank 0 (include/sim5gpu_rccl.h).  The ranks find each other through a file that rank 0 writes the
 * 128-byte communicator id to.  The file's name carries a per-run nonce -- SIM5_EXAMPLE_NONCE, or the parent process id,
 * which the ranks of one shell loop share -- and rank 0 removes it before it writes and when it exits, so a second run of
 * the same command never reads the id of the first.  Start the ranks before anything touches the GPU, e.g.
 *
 *     for r in 0 1 2 3 4 5 6 7; do HIP_VISIBLE_DEVICES=$r ./disk_image_sharded $r 8 /tmp/s5.id 0.998 70 4096 20 & done; wait
 *
 *   usage: disk_image_sharded <rank> <world> <id-file> [spin incl_deg n images]
 *   build: gcc -O2 -Iinclude examples/disk_image_sharded.c -o disk_image_sharded \
 *              -Lsim5_amd/lib -lsim5gpu_rccl -lsim5gpu -Wl,-rpath,$PWD/sim5_amd/lib -lm
 * Rank 0 prints the rate and the number of pixels that hit the disk (the reference's figure for the image).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "sim5gpu_rccl.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "ERROR: %s -> %d: %s | %s\n", #call, rc_, \
    sim5gpu_rccl_last_error(), sim5gpu_last_error()); return 1; } } while (0)

/* SIM5_EXAMPLE_VERBOSE=1: a line on stderr after every stage, each behind a synchronisation (to place a GPU fault) */
static int verbose = 0;
#define STAGE(what) do { if (verbose) { CHECK(sim5gpu_synchronize(NULL)); fprintf(stderr, "[rank %d] %s\n", rank, what); } } while (0)

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s rank world id-file [spin incl_deg n images]\n", argv[0]); return 2; }
    const int rank = atoi(argv[1]), world = atoi(argv[2]);
    char idfile[4096];
    {
        /* The id file carries a per-RUN stamp, the same on every rank: SIM5_EXAMPLE_NONCE (e.g. the job id, or $$ of the
         * launching shell).  With more than one rank it is REQUIRED: ranks started by mpirun / srun / xargs or on several nodes
         * have different parents, so a default derived from the process tree would make the peers wait for a file that never
         * comes, and a stamp that does not change between runs would hand a crashed run's id to the next one. */
        const char *nonce = getenv("SIM5_EXAMPLE_NONCE");
        if (world > 1 && !(nonce && *nonce)) {
            fprintf(stderr, "ERROR: set SIM5_EXAMPLE_NONCE to a value unique to this run and equal on all %d ranks (e.g. SIM5_EXAMPLE_NONCE=$$ in the launching shell)\n", world);
            return 2;
        }
        snprintf(idfile, sizeof idfile, "%s.%s", argv[3], (nonce && *nonce) ? nonce : "single");
    }
    const double a = argc > 4 ? atof(argv[4]) : 0.998, inc = (argc > 5 ? atof(argv[5]) : 70.0) / 180.0 * M_PI;
    const int n = argc > 6 ? atoi(argv[6]) : 4096, images = argc > 7 ? atoi(argv[7]) : 10;
    char id[SIM5GPU_RCCL_ID_BYTES];
    void *comm = NULL;
    verbose = getenv("SIM5_EXAMPLE_VERBOSE") != NULL;

    if (world > 1) {
        if (rank == 0) {                                  /* the id travels through a file: write, then rename into place */
            char tmp[4200];
            unlink(idfile);                               /* a file left by a run that died: never handed to this run's peers */
            CHECK(sim5gpu_rccl_unique_id(id));
            snprintf(tmp, sizeof tmp, "%s.tmp", idfile);
            FILE *f = fopen(tmp, "wb");
            if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) { perror(tmp); return 1; }
            fclose(f);
            if (rename(tmp, idfile) != 0) { perror(idfile); return 1; }
        } else {
            FILE *f = NULL;
            for (int tries = 0; tries < 600 && !(f = fopen(idfile, "rb")); tries++) usleep(100000);
            if (!f || fread(id, 1, sizeof id, f) != sizeof id) { fprintf(stderr, "ERROR: no communicator id in %s\n", idfile); return 1; }
            fclose(f);
        }
        CHECK(sim5gpu_rccl_comm_create(id, rank, world, &comm));
    }

    sim5gpu_image_desc img;
    memset(&img, 0, sizeof img);
    img.nx = n; img.ny = n; img.y0 = 0; img.y1 = n;
    img.a = a; img.incl = inc;
    img.bh_mass = 10.0; img.mdot = 0.1; img.alpha_visc = 0.1;         /* disk_nt_setup(10, a, 0.1, 0.1, 0): ref disk-image.c:45 */
    img.max_order = 2; img.disk_spin = -1.0;

    sim5gpu_shard *sh = NULL;
    CHECK(sim5gpu_shard_create(&sh, comm, rank, world, n, n, 0));
    int rows = 0, b0 = 0, b1 = 0;
    CHECK(sim5gpu_shard_plan(&img, rank, world, 0, &rows, &b0, &b1, NULL));
    float *d_f = NULL, *d_g = NULL;
    if (rank == 0) {
        CHECK(sim5gpu_malloc((void **)&d_f, (size_t)n * n * sizeof(float)));
        CHECK(sim5gpu_malloc((void **)&d_g, (size_t)n * n * sizeof(float)));
    }
    /* pipelined: the gather of image i runs while image i+1 is traced */
    STAGE("buffers allocated");
    CHECK(sim5gpu_disk_image_sharded(sh, &img, d_f, d_g, NULL));       /* warm-up: code, tables, first collective */
    CHECK(sim5gpu_synchronize(NULL));
    STAGE("warm-up image done");
    const double t0 = now();
    for (int i = 0; i < images; i++) {
        CHECK(sim5gpu_shard_image_begin(sh, &img, d_f, d_g, NULL));
        if (i > 0) CHECK(sim5gpu_shard_image_end(sh, NULL));
        STAGE("image begun");
    }
    if (images > 0) CHECK(sim5gpu_shard_image_end(sh, NULL));
    CHECK(sim5gpu_synchronize(NULL));
    const double dt = now() - t0;
    STAGE("timed images done");
    if (rank == 0) {
        float *g = (float *)malloc((size_t)n * n * sizeof(float));
        CHECK(sim5gpu_memcpy_d2h(g, d_g, (size_t)n * n * sizeof(float)));
        long hits = 0;
        for (size_t i = 0; i < (size_t)n * n; i++) hits += g[i] > 0.0f;
        printf("ranks %d  image %d x %d  rows on rank 0: %d  images %d  %.3f ms per image  %.4e rays/s  disk hits %ld\n",
               world, n, n, rows, images, images > 0 ? 1e3 * dt / images : 0.0, images > 0 ? (double)n * n * images / dt : 0.0, hits);
        free(g);
        fflush(stdout);
        sim5gpu_free(d_f); sim5gpu_free(d_g);
    }
    STAGE("image planes released");
    CHECK(sim5gpu_shard_destroy(sh));
    if (comm) CHECK(sim5gpu_rccl_comm_destroy(comm));
    if (world > 1 && rank == 0) unlink(idfile);           /* every peer has joined the communicator by now */
    return 0;
}
