// tools/probes/walk_sim.c — DIAGNOSTICS (CPU): counting copies of the matchfinder's class walk (zh_mf_frontier) on real windows, to price ideas
// before building them. For 64 KiB max-blocks with 32 KiB of history (every 7th of a corpus file): the K-gram classes in position order, then per position
//   current: the walk as built — candidates nearest first, a 4-byte probe at [cur-3, cur], survivors verified from byte 0 — candidates, survivors, false survivors,
//            records and bytes compared per position;
//   chain:   the LCP-chain alternative — with a[p] = LCP(p, its class predecessor) riding along the order, LCP(i, next) = min(LCP(i, this), a[this]) unless the two are
//            equal, in which case the strings are compared on from there — how often that equal case occurs and how many bytes it compares.
// Round 5, Python sources / mixed stream of configuration 4: 27.4 / 200 candidates per position, 1.24 / 8.5 survivors (0.14 / 6.9 false); the chain method meets the
// equal case 12.6 / 81 times per position: ten times the verifications it would save. Class keys of 7, 8, 10 bytes (-DKK=...): 22.5 / 19.8 / 10.7 candidates on the
// sources, 166 / 147 / 124 on the mixed stream.
// build: gcc -O2 [-DKK=7] -o walk_sim tools/probes/walk_sim.c ; run: ./walk_sim <corpus file> <bytes>   (corpus files: tests/corpus.py real_text / mixed_config4 .tofile)
#ifndef KK
#define KK 6
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
static const uint8_t *W; static uint32_t WN;
static int cmp6(const void*a,const void*b){uint32_t x=*(const uint32_t*)a,y=*(const uint32_t*)b;int c=memcmp(W+x,W+y,KK);if(c)return c;return x<y?-1:1;}
static uint32_t lcp(uint32_t i,uint32_t p,uint32_t from,uint32_t maxlen){uint32_t l=from;while(l<maxlen&&W[i+l]==W[p+l])l++;return l;}
int main(int argc,char**argv){
  FILE*f=fopen(argv[1],"rb");size_t total=atol(argv[2]);uint8_t*d=malloc(total);total=fread(d,1,total,f);fclose(f);
  uint64_t npos=0,cand=0,surv=0,falsesurv=0,recs=0,verbytes=0, eq=0,eqbytes=0,eqrec=0, neq=0, firstbytes=0, longfirst=0, chainrecs=0;
  uint32_t bs=65536;
  for(size_t b0=0;b0+bs<=total;b0+=bs*7){ // sample every 7th block
    size_t ws=b0>=32768?b0-32768:0; uint32_t prev=b0-ws; W=d+ws; WN=prev+bs; uint32_t M=WN-(KK-1);
    uint32_t*ord=malloc(M*4);for(uint32_t i=0;i<M;i++)ord[i]=i;qsort(ord,M,4,cmp6);
    uint16_t*a=calloc(WN,2);
    for(uint32_t j=1;j<M;j++){uint32_t i=ord[j],p=ord[j-1];if(memcmp(W+i,W+p,KK)==0){uint32_t ml=WN-i<258?WN-i:258;a[i]=lcp(i,p,0,ml);} }
    for(uint32_t j=0;j<M;j++){uint32_t i=ord[j];if(i<prev)continue; // runs excluded? keep all
      uint32_t f4=W[i]; int isrun=W[i+1]==f4&&W[i+2]==f4&&W[i+3]==f4&&W[i+4]==f4&&W[i+5]==f4; if(isrun)continue;
      npos++; uint32_t maxlen=WN-i<258?WN-i:258; uint32_t cur=5; // assume 3,4,5 handled
      // current method
      {uint32_t c=cur;for(int k=(int)j-1;k>=0;k--){uint32_t p=ord[k];if(memcmp(W+i,W+p,KK))break;if(i-p>32768)break;cand++;
         uint32_t fo=c-3; if(memcmp(W+i+fo,W+p+fo,4)==0){surv++;uint32_t l=lcp(i,p,0,maxlen);verbytes+=l; if(l>c){recs++;c=l;}else falsesurv++;} if(c>=maxlen)break;}}
      // chain method
      {uint32_t c=cur,m=0xffff,bb=a[i];for(int k=(int)j-1;k>=0;k--){uint32_t p=ord[k];if(memcmp(W+i,W+p,KK))break;if(i-p>32768)break;
         uint32_t mn; if(k==(int)j-1){mn=a[i]; firstbytes+=mn; if(mn>=80)longfirst++;}
         else if(m!=bb){mn=m<bb?m:bb;neq++;} else {mn=lcp(i,p,m,maxlen);eq++;eqbytes+=mn-m+1; if(mn>c)eqrec++;}
         if(mn>c){c=mn;chainrecs++;} m=mn;bb=a[p]; if(c>=maxlen)break;}}
    }
    free(ord);free(a);
  }
  printf("positions %lu\ncurrent: candidates/pos %.2f survivors/pos %.3f (false %.3f) records/pos %.3f verify-bytes/pos %.1f\n",npos,(double)cand/npos,(double)surv/npos,(double)falsesurv/npos,(double)recs/npos,(double)verbytes/npos);
  printf("chain: first-bytes/pos %.1f (first>=80: %.3f/pos) non-equal/pos %.2f equal/pos %.3f equal-bytes/pos %.2f equal-records/pos %.3f records/pos %.3f\n",(double)firstbytes/npos,(double)longfirst/npos,(double)neq/npos,(double)eq/npos,(double)eqbytes/npos,(double)eqrec/npos,(double)chainrecs/npos);
  return 0;}
