// host I/O microbenchmark: page-cache read / write throughput by method and thread count
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <thread>
#include <vector>
static double now(){return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();}
int main(int argc,char**argv){
  const char*dir=argc>1?argv[1]:"/tmp"; size_t GB=argc>2?atol(argv[2]):4; size_t N=GB<<30;
  char path[256]; snprintf(path,256,"%s/iob.dat",dir);
  char*src=(char*)aligned_alloc(4096,N); 
  {std::vector<std::thread> th; for(int t=0;t<32;++t) th.emplace_back([&,t]{size_t lo=N*t/32,hi=N*(t+1)/32; for(size_t i=lo;i<hi;i+=8) *(size_t*)(src+i)=i*0x9E3779B97F4A7C15ull;}); for(auto&x:th)x.join();}
  for(int T: {1,4,16,32,64}){
    // buffered pwrite, T threads, one file
    unlink(path); int fd=open(path,O_CREAT|O_RDWR|O_TRUNC,0644);
    double t0=now(); {std::vector<std::thread> th; for(int t=0;t<T;++t) th.emplace_back([&,t]{size_t lo=N*t/T,hi=N*(t+1)/T; const size_t CH=8<<20; for(size_t at=lo;at<hi;at+=CH){size_t n=std::min(CH,hi-at); if(pwrite(fd,src+at,n,at)!=(ssize_t)n) abort();}}); for(auto&x:th)x.join();}
    double dt=now()-t0; printf("pwrite 1 file  T=%2d: %.2f GB/s\n",T,N/dt/1e9); close(fd);
    // mmap write
    unlink(path); fd=open(path,O_CREAT|O_RDWR|O_TRUNC,0644); t0=now(); if(ftruncate(fd,N)) abort(); char*m=(char*)mmap(0,N,PROT_READ|PROT_WRITE,MAP_SHARED,fd,0);
    {std::vector<std::thread> th; for(int t=0;t<T;++t) th.emplace_back([&,t]{size_t lo=N*t/T,hi=N*(t+1)/T; memcpy(m+lo,src+lo,hi-lo);}); for(auto&x:th)x.join();}
    munmap(m,N); dt=now()-t0; printf("mmap write     T=%2d: %.2f GB/s\n",T,N/dt/1e9); close(fd);
    // pread (file is in page cache now) into src (pre-faulted)
    fd=open(path,O_RDONLY); t0=now(); {std::vector<std::thread> th; for(int t=0;t<T;++t) th.emplace_back([&,t]{size_t lo=N*t/T,hi=N*(t+1)/T; const size_t CH=8<<20; for(size_t at=lo;at<hi;at+=CH){size_t n=std::min(CH,hi-at); if(pread(fd,src+at,n,at)!=(ssize_t)n) abort();}}); for(auto&x:th)x.join();}
    dt=now()-t0; printf("pread          T=%2d: %.2f GB/s\n",T,N/dt/1e9);
    // mmap read + memchr count
    t0=now(); m=(char*)mmap(0,N,PROT_READ,MAP_SHARED,fd,0); std::vector<size_t> cnt(T);
    {std::vector<std::thread> th; for(int t=0;t<T;++t) th.emplace_back([&,t]{size_t lo=N*t/T,hi=N*(t+1)/T,c=0; const char*p=m+lo,*e=m+hi; while(p<e){const char*q=(const char*)memchr(p,'\n',e-p); if(!q)break; ++c; p=q+1;} cnt[t]=c;}); for(auto&x:th)x.join();}
    munmap(m,N); dt=now()-t0; printf("mmap read+memchr T=%2d: %.2f GB/s\n",T,N/dt/1e9); close(fd);
    // memcpy src->dst in memory
    fflush(stdout);
  }
  // two files, 16 threads each, buffered
  {char p2[256]; snprintf(p2,256,"%s/iob2.dat",dir); unlink(path); unlink(p2); int f1=open(path,O_CREAT|O_RDWR|O_TRUNC,0644),f2=open(p2,O_CREAT|O_RDWR|O_TRUNC,0644); double t0=now(); std::vector<std::thread> th; size_t H=N/2; for(int t=0;t<32;++t) th.emplace_back([&,t]{int fd=t&1?f2:f1; int tt=t>>1; size_t lo=H*tt/16,hi=H*(tt+1)/16; const size_t CH=8<<20; for(size_t at=lo;at<hi;at+=CH){size_t n=std::min(CH,hi-at); if(pwrite(fd,src+(t&1?H:0)+at,n,at)!=(ssize_t)n) abort();}}); for(auto&x:th)x.join(); double dt=now()-t0; printf("pwrite 2 files 16 thr each: %.2f GB/s total\n",N/dt/1e9); close(f1);close(f2); unlink(p2);}
  unlink(path); return 0;}
