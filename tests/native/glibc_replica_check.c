// Host check of the generated glibc 2.35 FMA-variant replicas (glibc235_fma_math.h) against the live libm.
// Built and run by tests/test_core_host.py.
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include "glibc235_fma_math.h"
static uint64_t s=88172645463325252ULL;
static inline uint64_t rnd(){ s^=s<<13; s^=s>>7; s^=s<<17; return s; }
static inline double u01(){ return (rnd()>>11)*(1.0/9007199254740992.0); }
int main(int argc,char**argv){
  long N = argc>1? atol(argv[1]):10000000;
  long bad[6]={0,0,0,0,0,0}; int o;
  double (*volatile pacos)(double)=acos; double (*volatile pasin)(double)=asin;
  double (*volatile psin)(double)=sin; double (*volatile pcos)(double)=cos;
  double (*volatile patan2)(double,double)=atan2; double (*volatile ppow)(double,double)=pow;
  for(long i=0;i<N;i++){
    double th=(u01()*2-1)*3.2; 
    if(i%7==0) th*=1e-3; if (i%11==0) th*=1e-6;
    double a=rpp_glibc_sin(th), b=psin(th);
    if(memcmp(&a,&b,8)){ if(bad[0]++<5) printf("sin %a: %a vs %a\n",th,a,b);}
    a=rpp_glibc_cos(th); b=pcos(th);
    if(memcmp(&a,&b,8)){ if(bad[1]++<5) printf("cos %a: %a vs %a\n",th,a,b);}
    double y=(u01()*2-1)*120, x=(u01()*2-1)*120;
    if(i%5==0){ y*=1e-3; } if(i%13==0){ x*=1e-4; } if(i%1000==0) x=0; if(i%1001==0) y=0;
    a=rpp_glibc_atan2(y,x); b=patan2(y,x);
    if(memcmp(&a,&b,8)){ if(bad[2]++<5) printf("atan2 %a %a: %a vs %a\n",y,x,a,b);}
    double p=(u01()*2-1)*150; if(i%9==0) p*=1e-5; if(i%997==0) p=0;
    a=rpp_glibc_pow(p,2.0); b=ppow(p,2.0);
    if(memcmp(&a,&b,8)){ if(bad[3]++<5) printf("pow %a: %a vs %a\n",p,a,b);}
    double ac=(u01()*2-1); if(i%17==0) ac*=1e-4; if(i%19==0) ac = ac>0? 1-ac*1e-6 : -1-ac*1e-6; if(i%997==1) ac=1; if(i%997==2) ac=-1;
    a=rpp_glibc_acos(ac); b=pacos(ac);
    if(memcmp(&a,&b,8)){ if(bad[4]++<5) printf("acos %a: %a vs %a\n",ac,a,b);}
    a=rpp_glibc_asin(ac); b=pasin(ac);
    if(memcmp(&a,&b,8)){ if(bad[5]++<5) printf("asin %a: %a vs %a\n",ac,a,b);}
  }
  printf("N=%ld mismatches sin=%ld cos=%ld atan2=%ld pow=%ld acos=%ld asin=%ld\n",N,bad[0],bad[1],bad[2],bad[3],bad[4],bad[5]);
  return (bad[0]||bad[1]||bad[2]||bad[3]||bad[4]||bad[5]);
}
