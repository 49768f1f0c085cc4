// lerp_q_check.c -- brute-force evidence for bq_device.hip.h: lerp_q (fp32 fma evaluation of the constant-weight
// lerps) against the contract (double-evaluated lerp): gcc -O2 -march=native -ffp-contract=off tools/lerp_q_check.c -lm
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
static inline float asf(uint32_t u){float f;memcpy(&f,&u,4);return f;}
static inline uint32_t asu(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static float ref(float a,float b,float c){ float cb=c*b; return (float)((1.0-(double)c)*(double)a+(double)cb);}   /* contract */
static float fast(float a,float b,float c){ float cb=c*b; return fmaf((float)(1.0-(double)c),a,cb);}
static uint64_t s=88172645463325252ull;
static uint64_t rnd(){s^=s<<13;s^=s>>7;s^=s<<17;return s;}
static int same(float x,float y){ if(x!=x&&y!=y)return 1; return x==y; }
int main(){
  float cs[3]={0.25f,0.5f,0.75f};
  for(int ci=0;ci<3;ci++){
    float c=cs[ci]; long bad=0,flagged=0,badunflag=0,N=0;
    for(long it=0;it<400000000L;it++){
      uint32_t ua=(uint32_t)rnd(), ub=(uint32_t)rnd();
      int mode=it&7;
      float a=asf(ua),b;
      if(mode<2) b=asf(ub);
      else if(mode<4){ /* b tiny relative to a: exponent gap 24..70 */
        int ea=(ua>>23)&255; int gap=20+(rnd()%60); int eb=ea-gap; if(eb<0)eb=0; b=asf((ub&0x807fffff)|((uint32_t)eb<<23)); }
      else if(mode<6){ int ea=(ua>>23)&255; int gap=(rnd()%8); int eb=ea-gap; if(eb<0)eb=0; b=asf((ub&0x807fffff)|((uint32_t)eb<<23)); }
      else { /* a tiny relative to b */ int eb=(ub>>23)&255; int gap=20+(rnd()%60); int ea=eb-gap; if(ea<0)ea=0; b=asf(ub); a=asf((ua&0x807fffff)|((uint32_t)ea<<23)); }
      float r=ref(a,b,c), f=fast(a,b,c);
      float cb=c*b; float q=fabsf(cb)*0x1p50f; int flag=(q>0.f)&&(q<fabsf(a));
      N++; if(flag)flagged++;
      if(!same(r,f)){bad++; if(!flag){badunflag++; if(badunflag<5)printf("c=%g a=%a b=%a ref=%a fast=%a\n",c,a,b,r,f);}}
    }
    printf("c=%g N=%ld mismatches=%ld flagged=%ld mismatches_unflagged=%ld\n",c,N,bad,flagged,badunflag);
  }
  return 0;
}
