#!/bin/bash
# development helper: where a wave's cycles and lanes go.  Needs variants/libprt_prof.so (tools/build_variant.sh prof -DPT_PROFILE
# -DPT_DEV_ONE_VARIANT).  usage: tools/prof_run.sh "ENV=val ..." ...   (cornell_diffuse 1080p, SPP from the environment, default 128)
P="photorealistic-rendering-using-opencl_amd"
for v in "$@"; do
  env PRT_LIB=$PWD/$P/variants/libprt_prof.so PRT_PROFILE=1 $v timeout -k 10 300 python3 bench.py --spp ${SPP:-128} --steps 1 --warmup 0 --no-cpu-baseline 2>&1 >/dev/null | grep PRT_PROFILE | python3 -c "
import sys
for l in sys.stdin:
    c=[int(x) for x in l.split()[1:]]
    it,bs,ds,laneA,go,lanestepsB,finC,laneC,laneE,lanes_end=c[0:10]
    cyc=c[10:17]; tot=c[17]; waves=c[18]; walkingB=c[20]; walkingD=c[21]; lanestepsD=c[22]; pre=c[19]
    segs=laneE + 0
    print('$v')
    print('  waves %d iterations/wave %.1f ; per iteration: B steps %.2f (lanes/step %.1f) D steps %.2f (lanes/step %.1f)' % (waves, it/waves, bs/it, lanestepsB/max(bs,1), ds/it, lanestepsD/max(ds,1)))
    print('  lanes per iteration: A %.1f  walkingB %.1f (unfinished at start %.1f)  closest_done %.1f  C %.1f  walkingD %.1f  E %.1f' % (laneA/it, walkingB/it, go/it, finC/it, laneC/it, walkingD/it, laneE/it))
    names=['A front','B begin','B loop','B done','C back','D walk','E finish']
    s=sum(cyc)+pre
    print('  cycles per iteration %.0f :' % (tot/it) + ' '.join('%s %.1f%%' % (n, 100.0*x/s) for n,x in zip(names,cyc)) + ' loop-head %.1f%%' % (100.0*pre/s))
"
done
