#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static uint64_t T[32];
static inline uint64_t asu64(double d){uint64_t u;memcpy(&u,&d,8);return u;}
static inline double asd(uint64_t u){double d;memcpy(&d,&u,8);return d;}
static inline uint32_t asu32(float f){uint32_t u;memcpy(&u,&f,4);return u;}
#define N 32
static const double InvLn2N = 0x1.71547652b82fep+0 * N;
static const double SHIFT = 0x1.8p+52;
static const double C0 = 0x1.c6af84b912394p-5/N/N/N, C1 = 0x1.ebfce50fac4f3p-3/N/N, C2 = 0x1.62e42ff0c52d6p-1/N;
static float emu(float x, int usefma){
  uint32_t ix = asu32(x); uint32_t abstop = (ix>>20)&0x7ff;
  if (abstop >= (asu32(88.0f)>>20)) {
    if (ix == asu32(-INFINITY)) return 0.0f;
    if (abstop >= (asu32(INFINITY)>>20)) return x + x;
    if (x > 0x1.62e42ep6f) return INFINITY;
    if (x < -0x1.9fe368p6f) return 0.0f;
    if (x < -0x1.9d1d9ep6f) return 0x1.4p-75f * 0x1.4p-75f;
  }
  double xd = x, z = InvLn2N * xd;
  double kd = z + SHIFT; uint64_t ki = asu64(kd); kd -= SHIFT;
  double r = z - kd;
  uint64_t t = T[ki % N]; t += ki << (52 - 5);
  double s = asd(t);
  double y;
  if (usefma) { z = fma(C0, r, C1); double r2 = r*r; y = fma(C2, r, 1.0); y = fma(z, r2, y); }
  else { z = C0*r + C1; double r2 = r*r; y = C2*r + 1; y = z*r2 + y; }
  y = y * s;
  return (float)y;
}
int main(){
  for (int i=0;i<32;i++){ long double v = powl(2.0L, (long double)i/32.0L); double d=(double)v; T[i]=asu64(d) - ((uint64_t)i<<47); }
  printf("T[1]=%016lx T[2]=%016lx T[31]=%016lx\n", T[1],T[2],T[31]);
  long bad_f=0,bad_n=0,n=0; srand(1);
  for (long it=0; it<60000000; ++it){
    float x;
    if (it < 20000000) x = -((float)rand()/RAND_MAX)*40.0f;
    else if (it < 40000000) x = ((float)rand()/RAND_MAX-0.5f)*200.0f;
    else { uint32_t u = ((uint32_t)rand()<<16) ^ (uint32_t)rand(); memcpy(&x,&u,4); if (x!=x) continue; }
    float a = expf(x); float b = emu(x,1), c = emu(x,0);
    if (asu32(a)!=asu32(b)) { if (bad_f<5) printf("fma mismatch x=%a libm=%a emu=%a\n",x,a,b); bad_f++; }
    if (asu32(a)!=asu32(c)) bad_n++;
    n++;
  }
  printf("n=%ld mismatches: fma-variant %ld, nofma-variant %ld\n", n,bad_f,bad_n);
}
