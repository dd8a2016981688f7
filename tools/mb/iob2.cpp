// host I/O microbenchmark 2: does preallocation or O_DIRECT lift the ~9 GB/s of buffered writes to one file?
#include <fcntl.h>
#include <unistd.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <thread>
#include <vector>
#include <algorithm>
static double now(){return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();}
int main(int argc,char**argv){
  const char*dir=argc>1?argv[1]:"/tmp"; size_t GB=argc>2?atol(argv[2]):4; size_t N=GB<<30;
  char path[256]; snprintf(path,256,"%s/iob.dat",dir);
  char*src=(char*)aligned_alloc(4096,N);
  {std::vector<std::thread> th; for(int t=0;t<32;++t) th.emplace_back([&,t]{size_t lo=N*t/32,hi=N*(t+1)/32; for(size_t i=lo;i<hi;i+=8) *(size_t*)(src+i)=i*0x9E3779B97F4A7C15ull;}); for(auto&x:th)x.join();}
  auto run=[&](const char*name,int flags,bool prealloc,int T,size_t CH){
    unlink(path); int fd=open(path,O_CREAT|O_RDWR|O_TRUNC|flags,0644); if(fd<0){printf("%s: open failed\n",name);return;}
    double t0=now(); if(prealloc && posix_fallocate(fd,0,N)!=0){printf("%s: fallocate failed\n",name);close(fd);return;}
    double t1=now(); bool ok=true;
    {std::vector<std::thread> th; for(int t=0;t<T;++t) th.emplace_back([&,t]{size_t lo=N/T*t,hi=t==T-1?N:N/T*(t+1); for(size_t at=lo;at<hi;at+=CH){size_t n=std::min(CH,hi-at); if(pwrite(fd,src+at,n,at)!=(ssize_t)n){ok=false;return;}}}); for(auto&x:th)x.join();}
    double dt=now()-t1; printf("%-34s T=%2d chunk %3zu MB: %s %.2f GB/s (prealloc %.2f s)\n",name,T,CH>>20,ok?"":"FAILED",N/dt/1e9,t1-t0); fflush(stdout); close(fd);};
  run("buffered",0,false,1,8<<20);
  run("buffered, 64 MB writes",0,false,1,64<<20);
  run("buffered + fallocate",0,true,1,8<<20);
  run("buffered + fallocate",0,true,8,8<<20);
  run("O_DIRECT + fallocate",O_DIRECT,true,1,8<<20);
  run("O_DIRECT + fallocate",O_DIRECT,true,8,8<<20);
  run("O_DIRECT + fallocate",O_DIRECT,true,32,8<<20);
  run("O_DIRECT",O_DIRECT,false,8,8<<20);
  unlink(path);
  FILE*fp=popen("df -T /tmp | tail -1; grep -E ' /tmp | / ' /proc/mounts | head -3","r"); char b[512]; while(fp&&fgets(b,512,fp)) fputs(b,stdout); if(fp)pclose(fp);
  return 0;}
