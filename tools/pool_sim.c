// tools/pool_sim.c -- development aid, companion of tools/walk_sim.c: prices a WORKGROUP-level ray pool on a walk trace
// (tools/walk_trace.py).  A workgroup is NW waves (adjacent 8x8 tiles).  A lane's closest-hit walk is taken in its own wave for
// at most S steps beyond the root step; a ray that needs more goes into the workgroup's pool and its lane waits.  A wave that
// finds >= FILL rays in the pool after an iteration (or has no runnable lane left) walks up to 64 of them in lock step to the
// end and hands the results back; the owners go on in their next iteration.  Waves run on their own clocks (vector instructions
// issued); a wave with nothing to do waits for the next result.  Output: vector instructions per wave-segment (throughput) and
// the makespan of the workgroup against the mean clock of its waves (how much of the wave slots' time is waiting).
//   gcc -O2 -o /tmp/pool_sim tools/pool_sim.c && /tmp/pool_sim <NW> <S> <FILL> <NF> [restart=1] [R=1]
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
static uint8_t* tr; static int H = 1080, W = 1920, F = 40;
#define TR(y, x, f, k) tr[((((size_t)(y) * W + (x)) * F + (f)) * 4) + (k)]
static double VA = 310, VCAM = 220, VB = 340, VC = 530, VE = 60, CSTEP = 61, VPOOL = 80, VPUSH = 30;
enum { READY = 0, WALKC = 1, SHADOW = 2, BACK = 4, FIN = 5, PARKED = 6, GOT = 7 };
typedef struct { int f, state, rem, kind, pendcam; double ready_at; } Lane;
typedef struct { int wave, lane, rem; } PoolRay;
#define MAXW 8
int main(int argc, char** argv) {
    int NW = atoi(argv[1]), S = atoi(argv[2]), FILL = atoi(argv[3]), NF = atoi(argv[4]);
    int restart = argc > 5 ? atoi(argv[5]) : 1;
    int R = argc > 6 ? atoi(argv[6]) : 1;      // a wave starts an iteration only with >= R runnable lanes (else it serves the pool or waits)
    const int rootfree = 1;
    FILE* fp = fopen("/tmp/walk_trace_cornell_diffuse.npy", "rb");
    fseek(fp, 0, SEEK_END); long sz = ftell(fp); long hdr = sz - (long)H * W * F * 4; fseek(fp, hdr, SEEK_SET);
    tr = malloc((size_t)H * W * F * 4);
    if (fread(tr, 1, (size_t)H * W * F * 4, fp) != (size_t)H * W * F * 4) return 1;
    double valu = 0, segs = 0, iters = 0, nwaves = 0, makespan = 0, clocksum = 0, poolwalks = 0, poollanes = 0, poolsteps = 0, waitcost = 0;
    for (int ty = 0; ty < H / 8; ty += 4) for (int tx = 0; tx + NW <= W / 8; tx += 4 * NW > 16 ? 4 * NW : 16) {
        static Lane L[MAXW][64]; int px[MAXW][64], py[MAXW][64]; double clk[MAXW];
        static PoolRay pool[MAXW * 64]; int npool = 0;
        for (int w = 0; w < NW; w++) { clk[w] = 0; nwaves++;
            for (int l = 0; l < 64; l++) { Lane z = {0, READY, 0, 0, 0, 0.0}; L[w][l] = z; px[w][l] = (tx + w) * 8 + (l & 7); py[w][l] = ty * 8 + (l >> 3); } }
#define FR(w, l) (8 + (L[w][l].f % 32))
        for (;;) {
            // the wave with the smallest clock that has something to do runs next
            int w = -1; double best = 1e300; int anyalive = 0;
            for (int k = 0; k < NW; k++) {
                int alive = 0; for (int l = 0; l < 64; l++) if (L[k][l].f < NF || L[k][l].state != READY) alive++;
                if (!alive) continue; anyalive = 1;
                if (clk[k] < best) { best = clk[k]; w = k; }
            }
            if (!anyalive) break;
            // results that have arrived
            int runnable = 0, parked = 0; double next_ready = 1e300;
            for (int l = 0; l < 64; l++) {
                Lane* a = &L[w][l];
                if (a->state == GOT) { if (a->ready_at <= clk[w]) { a->state = WALKC; a->rem = 0; } else if (a->ready_at < next_ready) next_ready = a->ready_at; }
                if (a->state == PARKED) parked++;
                if ((a->state == READY && a->f < NF) || a->state == WALKC) runnable++;
            }
            if (runnable < R && runnable > 0) {
                if (npool > 0) goto serve;
                if (next_ready < 1e300) { waitcost += next_ready - clk[w]; clk[w] = next_ready; continue; }
            }
            if (!runnable) {
                // serve the pool if it holds anything, else wait
                if (npool > 0) goto serve;
                if (next_ready < 1e300) { waitcost += next_ready - clk[w]; clk[w] = next_ready; continue; }
                // parked lanes whose rays sit in nobody's hands cannot happen: npool == 0 means every parked ray was served
                // by a wave whose clock has not reached the hand-over yet; let that wave run
                { double mn = 1e300; for (int k = 0; k < NW; k++) if (k != w && clk[k] > clk[w] && clk[k] < mn) mn = clk[k];
                  if (mn < 1e300) { waitcost += mn - clk[w]; clk[w] = mn; continue; } }
                fprintf(stderr, "stuck\n"); return 2;
            }
            iters++;
            {
                double c = 0; int nready = 0, ncam = 0;
                for (int l = 0; l < 64; l++) { Lane* a = &L[w][l]; if (a->state == READY && a->f < NF) {
                    int f = FR(w, l); int w1 = TR(py[w][l], px[w][l], f, 0), w2 = TR(py[w][l], px[w][l], f, 1);
                    if (w1 != 255 && !a->pendcam) { a->state = WALKC; a->rem = w1 > rootfree ? w1 - rootfree : 0; a->kind = 1; a->ready_at = -1; ncam++; continue; }
                    a->pendcam = 0; nready++;
                    if (w2 != 255) { a->state = WALKC; a->rem = w2 > rootfree ? w2 - rootfree : 0; a->kind = 0; a->ready_at = -1; }
                    else { segs++; a->f++; }
                } }
                if (nready) c += VA; if (ncam) c += VCAM;
                // B: in-wave walk of at most S steps; the rest is pooled
                int steps = 0, npush = 0;
                for (int l = 0; l < 64; l++) { Lane* a = &L[w][l]; if (a->state == WALKC && a->rem > 0) { int s = a->rem < S ? a->rem : S; if (s > steps) steps = s; } }
                c += steps * CSTEP;
                for (int l = 0; l < 64; l++) { Lane* a = &L[w][l]; if (a->state == WALKC && a->rem > 0) {
                    if (a->rem <= S) a->rem = 0;
                    else { if (!restart) a->rem -= S; pool[npool].wave = w; pool[npool].lane = l; pool[npool].rem = a->rem; npool++; a->state = PARKED; npush++; }
                } }
                if (npush) c += VPUSH;
                int nfin = 0, nprobe = 0;
                for (int l = 0; l < 64; l++) { Lane* a = &L[w][l]; if (a->state == WALKC && a->rem == 0) { nfin++; if (a->kind == 1) { a->state = READY; a->pendcam = 1; } else { nprobe++; a->state = BACK; } } }
                if (nfin) c += VB; if (nprobe) c += VC;
                steps = 0;
                for (int l = 0; l < 64; l++) { Lane* a = &L[w][l]; if (a->state == BACK) { int w3 = TR(py[w][l], px[w][l], FR(w, l), 2);
                    if (w3 != 255) { a->state = SHADOW; a->rem = w3 > rootfree ? w3 - rootfree : 0; if (a->rem > steps) steps = a->rem; } else a->state = FIN; } }
                c += steps * CSTEP;
                int nE = 0; for (int l = 0; l < 64; l++) { Lane* a = &L[w][l]; if (a->state == FIN || a->state == SHADOW) { nE++; segs++; a->state = READY; a->f++; } }
                if (nE) c += VE;
                clk[w] += c; valu += c;
            }
            if (npool < FILL) continue;
        serve: {
                int n = npool < 64 ? npool : 64; int mx = 0;
                for (int k = 0; k < n; k++) if (pool[k].rem > mx) mx = pool[k].rem;
                double c = VPOOL + mx * CSTEP;
                clk[w] += c; valu += c; poolwalks++; poollanes += n; poolsteps += mx;
                for (int k = 0; k < n; k++) { Lane* a = &L[pool[k].wave][pool[k].lane]; a->state = GOT; a->ready_at = clk[w]; }
                memmove(pool, pool + n, sizeof(PoolRay) * (npool - n)); npool -= n;
            }
        }
        double mk = 0; for (int w = 0; w < NW; w++) { if (clk[w] > mk) mk = clk[w]; clocksum += clk[w]; } makespan += mk * NW;
    }
    printf("NW=%d S=%d FILL=%d NF=%d restart=%d R=%d: VALU/wave-seg %.0f  iters/wave %.1f  pool walks/wave %.1f (lanes %.1f, steps %.1f)  wait share %.3f  makespan/mean clock %.3f\n",
           NW, S, FILL, NF, restart, R, valu / (segs / 64), iters / nwaves, poolwalks / nwaves, poollanes / (poolwalks > 0 ? poolwalks : 1), poolsteps / (poolwalks > 0 ? poolwalks : 1),
           waitcost / clocksum, makespan / clocksum);
    return 0;
}
