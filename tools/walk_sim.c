// tools/walk_sim.c -- development aid: prices wave schedules of the lane machine on a walk trace (tools/walk_trace.py) before
// they are written as kernels.  Every 8x8 tile of the trace is one wave; each lane replays its pixel's segments (the steps of
// W1 / W2 / W3 from the trace) through the iteration A, B, C, D, E of render_kernel until ALL lanes have done NF frames (the
// trace is repeated cyclically), with a walk phase that ends below T walking lanes (slack >= 0: never cut off a lane that is
// within `slack` frames of the slowest one).  Output: vector instructions per wave-segment under a fixed cost per phase
// (VA.. below, from the instruction counts of the kernel), iterations per wave, steps and lanes per step.
//   gcc -O2 -o /tmp/walk_sim tools/walk_sim.c && /tmp/walk_sim <T> <slack|-1> <NF>     (reads /tmp/walk_trace_cornell_diffuse.npy)
// What it taught: a simulation that stops before the end of the launch promises -25 % at T = 8; including the
// tail it puts the optimum at T = 4 (-7 %), which is what the GPU measured.
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
static uint8_t* tr; static int H=1080, W=1920, F=40;
#define TR(y,x,f,k) tr[((((size_t)(y)*W+(x))*F+(f))*4)+(k)]
static double VA=310, VCAM=220, VB=340, VC=530, VE=60, CSTEP=61;
typedef struct { int f, state, rem, kind, pendcam; } Lane;
int main(int argc,char**argv){
  int T=atoi(argv[1]); int slack=atoi(argv[2]); int NF=atoi(argv[3]); int rootfree=1;
  FILE*fp=fopen("/tmp/walk_trace_cornell_diffuse.npy","rb"); fseek(fp,0,SEEK_END); long sz=ftell(fp); long hdr=sz-(long)H*W*F*4; fseek(fp,hdr,SEEK_SET);
  tr=malloc((size_t)H*W*F*4); if(fread(tr,1,(size_t)H*W*F*4,fp)!=(size_t)H*W*F*4) return 1;
  double valu=0, segs=0, iters=0, stepsC=0, stepsS=0, laneA=0, lanestep=0, waves=0;
  for(int ty=0;ty<H/8;ty+=4) for(int tx=0;tx<W/8;tx+=4){
    Lane L[64]; int px[64],py[64]; waves++;
    for(int l=0;l<64;l++){ L[l].f=0; L[l].state=0; L[l].rem=0; L[l].kind=0; L[l].pendcam=0; px[l]=tx*8+(l&7); py[l]=ty*8+(l>>3);}
    #define FR(l) (8+(L[l].f%32))
    for(;;){
      int alive=0; for(int l=0;l<64;l++) if(L[l].f<NF || L[l].state!=0) alive++; if(!alive) break;
      iters++;
      int fmin=1<<30; for(int l=0;l<64;l++) if(L[l].f<NF||L[l].state!=0) if(L[l].f<fmin) fmin=L[l].f;
      int nready=0, ncam=0;
      for(int l=0;l<64;l++) if(L[l].state==0 && L[l].f<NF){
        int f=FR(l); int w1=TR(py[l],px[l],f,0), w2=TR(py[l],px[l],f,1);
        if(w1!=255 && !L[l].pendcam){ L[l].state=1; L[l].rem=w1>rootfree?w1-rootfree:0; L[l].kind=1; ncam++; continue; }
        L[l].pendcam=0; nready++;
        if(w2!=255){ L[l].state=1; L[l].rem=w2>rootfree?w2-rootfree:0; L[l].kind=0; }
        else { segs++; L[l].f++; }
      }
      if(nready){ valu+=VA; laneA+=nready; } if(ncam) valu+=VCAM;
      // B
      int steps=0, nstart=0, other=0; for(int l=0;l<64;l++) if(L[l].state==1){ if(L[l].rem>0) nstart++; else other=1; }
      for(;;){ int act=0, crit=0; for(int l=0;l<64;l++) if(L[l].state==1 && L[l].rem>0){ act++; if(L[l].f<=fmin+slack) crit++; }
        if(act==0) break;
        if(steps>0 && act<T && (slack<0 || crit==0) && (act<nstart||other)) break;
        steps++; lanestep+=act; for(int l=0;l<64;l++) if(L[l].state==1&&L[l].rem>0) L[l].rem--; }
      valu+=steps*CSTEP; stepsC+=steps;
      int nfin=0,nprobe=0;
      for(int l=0;l<64;l++) if(L[l].state==1 && L[l].rem==0){ nfin++; if(L[l].kind==1){ L[l].state=0; L[l].pendcam=1; } else { nprobe++; L[l].state=4; } }
      if(nfin) valu+=VB; if(nprobe) valu+=VC;
      for(int l=0;l<64;l++) if(L[l].state==4){ int w3=TR(py[l],px[l],FR(l),2); if(w3!=255){ L[l].state=2; L[l].rem=w3>rootfree?w3-rootfree:0; } else L[l].state=5; }
      steps=0; for(;;){ int act=0; for(int l=0;l<64;l++) if(L[l].state==2 && L[l].rem>0) act++; if(act==0) break; steps++; for(int l=0;l<64;l++) if(L[l].state==2&&L[l].rem>0) L[l].rem--; }
      valu+=steps*CSTEP; stepsS+=steps;
      int nE=0; for(int l=0;l<64;l++) if(L[l].state==5 || (L[l].state==2 && L[l].rem==0)){ nE++; segs++; L[l].state=0; L[l].f++; }
      if(nE) valu+=VE;
    }
  }
  printf("T=%2d slack=%2d NF=%d: VALU/wave-seg %.0f  iters/wave %.1f  closest steps/iter %.2f (lanes/step %.1f) shadow %.2f  A lanes %.1f\n",T,slack,NF, valu/(segs/64), iters/waves, stepsC/iters, lanestep/stepsC, stepsS/iters, laneA/iters);
  return 0; }
