// Stage rates of the exact-mode plan pipeline in isolation (generator, conversion, shuffle scan, swaps): 
// clang++ -O3 -std=c++17 -ffp-contract=off -pthread tools/ubench/mt_stages.cpp -o /tmp/mt_stages && /tmp/mt_stages
#include "../../emcee_amd/csrc/emx_mtpipe.cpp"
#include <cstdio>
using namespace emx;
int main(){
  std::vector<uint32_t> a(624,12345u), b(624), o(624);
  for(int i=0;i<624;i++) a[i]=i*2654435761u;
  auto t0=std::chrono::steady_clock::now();
  const int R=200000;
  for(int r=0;r<R;r+=2){ twist_block(a.data(),b.data()); temper_block(b.data(),o.data()); twist_block(b.data(),a.data()); temper_block(a.data(),o.data()); }
  double dt=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
  printf("gen: %.3f ns/word (%u)\n", dt*1e9/(R*624.0), o[5]);
  {
    TwistFn f = pick_twist();
    std::vector<uint32_t> ka(640, 1u), kb(640), oo(640);
    for (int i = 0; i < 624; i++) ka[i] = i * 2654435761u;
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < R; r += 2) { f(ka.data(), kb.data(), oo.data()); f(kb.data(), ka.data(), oo.data()); }
    dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("gen (%s): %.3f ns/word (%u)\n", f == twist_temper_generic ? "generic" : "avx512 register window", dt * 1e9 / (R * 624.0), oo[5]);
  }
  // tokenizer-like loops over a big static buffer
  std::vector<uint32_t> w(1<<22); for(size_t i=0;i<w.size();i++) w[i]=(uint32_t)(i*2654435761u ^ (i>>3)*40503u);
  std::vector<double> d(1<<21);
  t0=std::chrono::steady_clock::now();
  for(int r=0;r<20;r++) convert_pairs(w.data(), d.data(), 1<<21);
  dt=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
  printf("convert: %.3f ns/word\n", dt*1e9/(20.0*(1<<22)));
  // shuffle scan
  std::vector<uint32_t> j(65536);
  t0=std::chrono::steady_clock::now();
  size_t used=0;
  for(int r=0;r<200;r++){
    int64_t i=65535; const uint32_t* p=w.data()+r*1000;
    while(i>0){ uint32_t mask=(uint32_t)i; mask|=mask>>1;mask|=mask>>2;mask|=mask>>4;mask|=mask>>8;mask|=mask>>16; int64_t lo=mask>>1;
      while(i>lo){ uint32_t v=*p++&mask; j[i]=v; i-=(int64_t)(v<=(uint32_t)i);} }
    used+=p-(w.data()+r*1000);
  }
  dt=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
  printf("shuffle scan: %.3f ns/word, %.1f us per 65536 (%zu words)\n", dt*1e9/used, dt*1e6/200, used/200);
  // swaps
  std::vector<uint8_t> x(65536);
  t0=std::chrono::steady_clock::now();
  for(int r=0;r<200;r++){ for(int i=0;i<65536;i++) x[i]=i&1; for(int64_t i=65535;i>0;--i){uint32_t jj=j[i]; uint8_t t=x[i]; x[i]=x[jj]; x[jj]=t;} }
  dt=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
  printf("swaps: %.1f us per 65536 (%d)\n", dt*1e6/200, x[77]);
}
