// How often, and by how much, do candidate float64 sin / cos implementations differ from THIS host's libm (glibc - what NumPy calls in the
// reference) on the tick's argument rad = (yaw*pi)/180, yaw uniform in +-7500 degrees?  The experiment behind sincos_yaw in
// q1physrl_amd/csrc/q1env_device.hpp (variant B is what the kernels use; C and D drop the lo part / the compensated cosine and were
// rejected: 15.6 % / 25.3 % of results differ instead of 3.1 %).  CPU only; the device-side check is q1env_selftest_trig.
// Build + run: gcc -O2 -mfma -ffp-contract=off tools/sincos_accuracy.c -o /tmp/sincos_accuracy -lm && /tmp/sincos_accuracy
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
static const double S1=-1.66666666666666324348e-01,S2=8.33333333332248946124e-03,S3=-1.98412698298579493134e-04,S4=2.75573137070700676789e-06,S5=-2.50507602534068634195e-08,S6=1.58969099521155010221e-10;
static const double C1=4.16666666666666019037e-02,C2=-1.38888888888741095749e-03,C3=2.48015872894767294178e-05,C4=-2.75573143513906633035e-07,C5=2.08757232129817482790e-09,C6=-1.13596475577881948265e-11;
static const double P1=1.5707963267948966, P2=6.123233995736766e-17, P3=-1.4973849048591698e-33, TWO_OVER_PI=0.6366197723675814;
// B: hi/lo reduction (6 ops) + fdlibm kernels with lo
static void sc_B(double x, double*s, double*c){
  double n=rint(x*TWO_OVER_PI);
  double r0=fma(-n,P1,x);
  double y0=fma(-n,P2,r0);
  double t=r0-y0;
  double y1=fma(-n,P2,t);
  y1=fma(-n,P3,y1);
  double z=y0*y0, v=z*y0;
  double rs=fma(z,fma(z,fma(z,fma(z,S6,S5),S4),S3),S2);
  double sn=y0-((z*(0.5*y1-v*rs)-y1)-v*S1);
  double rc=z*fma(z,fma(z,fma(z,fma(z,fma(z,C6,C5),C4),C3),C2),C1);
  double hz=0.5*z, w=1.0-hz;
  double cs=w+(((1.0-w)-hz)+(z*rc-y0*y1));
  int q=(int)n;
  double ss=(q&1)?cs:sn, cc=(q&1)?sn:cs;
  if(q&2) ss=-ss; if((q+1)&2) cc=-cc;
  *s=ss;*c=cc;
}
// C: no lo
static void sc_C(double x, double*s, double*c){
  double n=rint(x*TWO_OVER_PI);
  double y0=fma(-n,P3,fma(-n,P2,fma(-n,P1,x)));
  double z=y0*y0, v=z*y0;
  double rs=fma(z,fma(z,fma(z,fma(z,fma(z,S6,S5),S4),S3),S2),S1);
  double sn=fma(v,rs,y0);
  double rc=fma(z,fma(z,fma(z,fma(z,fma(z,C6,C5),C4),C3),C2),C1);
  double hz=0.5*z, w=1.0-hz;
  double cs=w+(((1.0-w)-hz)+(z*z)*rc);
  int q=(int)n;
  double ss=(q&1)?cs:sn, cc=(q&1)?sn:cs;
  if(q&2) ss=-ss; if((q+1)&2) cc=-cc;
  *s=ss;*c=cc;
}
// D: like C but simplest cos
static void sc_D(double x, double*s, double*c){
  double n=rint(x*TWO_OVER_PI);
  double y0=fma(-n,P3,fma(-n,P2,fma(-n,P1,x)));
  double z=y0*y0, v=z*y0;
  double rs=fma(z,fma(z,fma(z,fma(z,fma(z,S6,S5),S4),S3),S2),S1);
  double sn=fma(v,rs,y0);
  double rc=fma(z,fma(z,fma(z,fma(z,fma(z,C6,C5),C4),C3),C2),C1);
  double cs=fma(z*z,rc,fma(z,-0.5,1.0));
  int q=(int)n;
  double ss=(q&1)?cs:sn, cc=(q&1)?sn:cs;
  if(q&2) ss=-ss; if((q+1)&2) cc=-cc;
  *s=ss;*c=cc;
}
static int64_t ulpdiff(double a,double b){int64_t x,y;memcpy(&x,&a,8);memcpy(&y,&b,8);if(x<0)x=INT64_MIN-x;if(y<0)y=INT64_MIN-y;return llabs(x-y);}
int main(){
  uint64_t st=88172645463325252ULL; long N=20000000;
  long dB=0,dC=0,dD=0; int64_t mB=0,mC=0,mD=0;
  for(long i=0;i<N;i++){
    st^=st<<13;st^=st>>7;st^=st<<17;
    double u=(st>>11)*(1.0/9007199254740992.0);
    double yaw=(u-0.5)*15000.0;
    double rad=(yaw*3.141592653589793)/180.0;
    double s0=sin(rad),c0=cos(rad),s,c;
    sc_B(rad,&s,&c); int64_t d=ulpdiff(s,s0)+0; int64_t e=ulpdiff(c,c0); dB+=(d!=0)+(e!=0); if(d>mB)mB=d; if(e>mB)mB=e;
    sc_C(rad,&s,&c); d=ulpdiff(s,s0); e=ulpdiff(c,c0); dC+=(d!=0)+(e!=0); if(d>mC)mC=d; if(e>mC)mC=e;
    sc_D(rad,&s,&c); d=ulpdiff(s,s0); e=ulpdiff(c,c0); dD+=(d!=0)+(e!=0); if(d>mD)mD=d; if(e>mD)mD=e;
  }
  printf("B differ %.4f%% max %ld ulp\nC differ %.4f%% max %ld\nD differ %.4f%% max %ld\n",100.0*dB/(2.0*N),(long)mB,100.0*dC/(2.0*N),(long)mC,100.0*dD/(2.0*N),(long)mD);
}
