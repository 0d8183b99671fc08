#include <stdio.h>
#include <math.h>
int main(){
  volatile float b=32767.f; const float c=1.0f/32767.f; int bad=0;
  for(int x=-32768;x<=32767;++x){ float a=(float)x; volatile float q=a/b; float q0=a*c; float r=fmaf(-32767.f,q0,a); float q1=fmaf(r,c,q0);
    if(q1!=q || (signbit(q1)!=signbit((float)q))) {bad++; if(bad<5) printf("x=%d %a %a\n",x,(double)q,(double)q1);} }
  printf("bad=%d\n",bad);
  volatile float b2=127.f; const float c2=1.0f/127.f; bad=0;
  for(int x=-128;x<=127;++x){ float a=(float)x; volatile float q=a/b2; float q0=a*c2; float r=fmaf(-127.f,q0,a); float q1=fmaf(r,c2,q0); if(q1!=q) bad++; }
  printf("bad127=%d\n",bad); return 0; }
