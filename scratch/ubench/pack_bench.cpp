#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "../../include/rufus_hip.h"
int main(int argc,char**argv){
  int fd=open(argv[1],O_RDONLY); struct stat st; fstat(fd,&st);
  const char* m=(const char*)mmap(0,st.st_size,PROT_READ,MAP_PRIVATE,fd,0);
  auto now=[]{return std::chrono::steady_clock::now();};
  auto t0=now(); unsigned long sum=0; for(size_t i=0;i<(size_t)st.st_size;i+=4096) sum+=m[i];
  auto t1=now(); printf("touch pages: %.3f s\n",std::chrono::duration<double>(t1-t0).count());
  std::vector<uint64_t> start; std::vector<uint32_t> slen;
  const char*p=m,*e=m+st.st_size;
  while(p<e){ const char*nl=(const char*)memchr(p,'\n',e-p); const char*s=nl+1; nl=(const char*)memchr(s,'\n',e-s); size_t L=nl-s; const char*pl=nl+1; nl=(const char*)memchr(pl,'\n',e-pl); const char*q=nl+1; const char*qe=(const char*)memchr(q,'\n',e-q); if(!qe)qe=e; start.push_back(s-m); slen.push_back(L); p=qe<e?qe+1:e;}
  auto t2=now(); printf("split lines: %.3f s (%zu reads)\n",std::chrono::duration<double>(t2-t1).count(),start.size());
  size_t n=start.size(); std::vector<uint64_t> codes(n*5); std::vector<uint32_t> acgt(n*5), wo(n+1), len(n);
  auto t3=now(); printf("alloc: %.3f s\n",std::chrono::duration<double>(t3-t2).count());
  for(int rep=0;rep<2;++rep){ wo[0]=0; auto a=now(); rfx_pack_spans(m,start.data(),slen.data(),nullptr,n,0,1,codes.data(),acgt.data(),nullptr,wo.data(),len.data()); auto b=now(); printf("pack: %.3f s\n",std::chrono::duration<double>(b-a).count());}
  return sum==1;
}
