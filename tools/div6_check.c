// exhaustive check of philox.h div6(): q=x*RN(1/6); q+=fma(-6,q,x)*RN(1/6) equals x/6.0f for every float of magnitude 1e-30..16
// gcc -O2 -ffp-contract=off tools/div6_check.c -lm && ./a.out   ->  1740340036 checked, 0 bad
#include <stdio.h>
#include <math.h>
#include <string.h>
#include <stdint.h>
int main(){ const float y = 1.0f/6.0f; long bad=0, n=0;
 for (uint32_t b=0; b<0x7f800000u; b++){ float x; memcpy(&x,&b,4); if (x>16.f) break; if (x!=0 && x<1e-30f) continue;
   for (int sg=0; sg<2; sg++){ float a = sg? -x: x; float q=a*y; float r=fmaf(-6.0f,q,a); float q2=fmaf(r,y,q); if (q2 != a/6.0f) bad++; n++; } }
 printf("%ld checked, %ld bad\n", n, bad); return 0; }
