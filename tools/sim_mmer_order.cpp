#include <cstdio>
#include <cstdint>
#include <algorithm>
#include <random>
#include <unordered_map>
#include <vector>
static inline uint32_t mul24(uint32_t a, uint32_t b){ return (uint32_t)((uint64_t)(a&0xFFFFFF)*(b&0xFFFFFF)); }
static uint32_t rc12(uint32_t s){ uint32_t r=0; for(int b=0;b<12;++b) r |= (3u-((s>>(2*b))&3u))<<(2*(11-b)); return r; }
int main(){
  for (int variant=0; variant<5; ++variant){
  std::mt19937_64 rng(1);
  std::unordered_map<uint32_t,int> cnt;
  const uint32_t span=(3u<<16)*16u;
  // random genome walk to measure density too
  const int N=2000000;
  std::vector<uint8_t> g(N+64); for(auto&c:g) c=rng()&3;
  uint32_t prev=0; long changes=0, tot=0;
  for (int p=18;p<N;++p){
    uint32_t mz=0xFFFFFFFFu;
    for(int j=0;j<8;++j){ uint32_t sub=0; for(int b=0;b<12;++b) sub=(sub<<2)|g[p-18+j+b]; uint32_t r=rc12(sub); uint32_t h;
      switch(variant){
        case 0: h=mul24(std::min(sub,r),0x9E3779); break;
        case 1: h=mul24(sub,r); break;
        case 2: h=mul24(mul24(sub,r),0x9E3779); break;
        case 3: h=mul24(sub+r,0x9E3779); break;
        default: h=mul24(sub^0x5A5A5A,r^0x5A5A5A); break;
      }
      mz=std::min(mz,h);}
    if (mz!=prev) ++changes; prev=mz; ++tot;
    uint32_t hh=mul24(mz,0xC2B2AF); uint32_t off=(uint32_t)(((uint64_t)hh*span)>>32)&~15u; cnt[off>>4]++;
  }
  int mx=0; double s2=0; for(auto&kv:cnt){ mx=std::max(mx,kv.second); s2+=(double)kv.second*kv.second; }
  printf("variant %d: density %.4f distinct blocks %zu of %u, max %d, sum n^2/N = %.2f (uniform %.2f)\n", variant, (double)changes/tot, cnt.size(), 3u<<16, mx, s2/tot, (double)tot/(3u<<16)+1);
  }
}
