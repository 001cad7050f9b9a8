// tools/probes/walk_sim2.c — DIAGNOSTICS (CPU), round 6: prices the MULTI-DEPTH class walk asked for in VERDICT.md (round 5, item 1) before building it.
// For 64 KiB max-blocks with 32 KiB of history (every 7th block of a corpus file), run-interior positions excluded as in zh_mf_frontier:
//   one level (as built): the walk over the 6-gram class, nearest first, until the record reaches the maximum, the class head or 32 KiB;
//   two levels (K1 = 6, K2): the 6-walk ends at the first candidate that shares >= K2 bytes (that candidate is the nearest member of the K2-gram class: every record
//            after it lies in that class), then a walk over the K2-gram class (an order of its own: classes contiguous, ascending in position);
//   three levels (6, K2, K3) likewise.
// Reported per position: candidates met by each level; and what the wave-synchronous walk of the kernel pays — per 64-entry chunk of an order the LONGEST lane's
// candidates / 4 (ZH_MF_STEP) steps, summed over the orders and divided by the chunks of the 6-gram order (the one-level figure is the kernel's 8.6 steps per chunk).
// build: gcc -O2 -o walk_sim2 tools/probes/walk_sim2.c ; run: ./walk_sim2 <corpus file> <bytes> <K2> [K3]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
static const uint8_t *W; static uint32_t WN; static int KK;
static int cmpk(const void*a,const void*b){uint32_t x=*(const uint32_t*)a,y=*(const uint32_t*)b;int c=memcmp(W+x,W+y,KK);if(c)return c;return x<y?-1:1;}
static uint32_t lcp(uint32_t i,uint32_t p,uint32_t maxlen){uint32_t l=0;while(l<maxlen&&W[i+l]==W[p+l])l++;return l;}
// walk of the K-gram class order `ord` (M entries) for every block position: stop when the record reaches stop_at (or maxlen), at the head, or out of reach.
// start_rec: the record the walk starts with. counts candidates per position into cand[], returns nothing; steps: sum over 64-entry chunks of ceil(max lane candidates / 4)
static void walk(const uint32_t*ord,uint32_t M,int K,uint32_t prev,uint32_t stop_at,uint64_t*cands,uint64_t*steps,uint64_t*chunks,uint64_t*recs,uint64_t*alive_lanesteps){
  for(uint32_t c=0;c<M;c+=64){ uint32_t mx=0; int any=0; uint32_t lanec[64]; uint32_t nl=0;
    for(uint32_t j=c;j<c+64&&j<M;j++){uint32_t i=ord[j]; if(i<prev)continue;
      uint32_t f4=W[i]; int isrun=W[i+1]==f4&&W[i+2]==f4&&W[i+3]==f4&&W[i+4]==f4&&W[i+5]==f4; if(isrun)continue;
      any=1; uint32_t maxlen=WN-i<258?WN-i:258; uint32_t cur=K-1,n=0; uint32_t lim=stop_at<maxlen?stop_at:maxlen;
      for(int k=(int)j-1;k>=0;k--){uint32_t p=ord[k]; if(memcmp(W+i,W+p,K))break; if(i-p>32768)break; n++;
        uint32_t fo=cur>=3?cur-3:0; if(memcmp(W+i+fo,W+p+fo,4)==0){uint32_t l=lcp(i,p,maxlen); if(l>cur){cur=l;(*recs)++;}} if(cur>=lim)break;}
      *cands+=n; if(n>mx)mx=n; lanec[nl++]=n; }
    if(any){(*chunks)++; uint32_t st=(mx+3)/4; *steps+=st; for(uint32_t s=0;s<st;s++){for(uint32_t l=0;l<nl;l++) if(lanec[l]>4*s)(*alive_lanesteps)++;}}
  }
}
int main(int argc,char**argv){
  FILE*f=fopen(argv[1],"rb");size_t total=atol(argv[2]);int K2=argc>3?atoi(argv[3]):12,K3=argc>4?atoi(argv[4]):0;
  uint8_t*d=malloc(total+300);total=fread(d,1,total,f);fclose(f);
  uint64_t npos=0,c1=0,s1=0,ch1=0,r1=0,a1=0, c6=0,s6=0,ch6=0,r6=0,a6=0, c2=0,s2=0,ch2=0,r2=0,a2=0, c3=0,s3=0,ch3=0,r3=0,a3=0, m2=0,m3=0;
  uint32_t bs=65536;
  for(size_t b0=0;b0+bs<=total;b0+=bs*7){
    size_t ws=b0>=32768?b0-32768:0; uint32_t prev=b0-ws; W=d+ws; WN=prev+bs;
    for(uint32_t i=prev;i<WN-5;i++){uint32_t f4=W[i];int isrun=W[i+1]==f4&&W[i+2]==f4&&W[i+3]==f4&&W[i+4]==f4&&W[i+5]==f4;if(!isrun)npos++;}
    uint32_t M=WN-5; uint32_t*ord=malloc(M*4);for(uint32_t i=0;i<M;i++)ord[i]=i; KK=6;qsort(ord,M,4,cmpk);
    walk(ord,M,6,prev,258,&c1,&s1,&ch1,&r1,&a1);            // one level
    walk(ord,M,6,prev,K2,&c6,&s6,&ch6,&r6,&a6);             // level 1 of two: ends once the record is >= K2
    free(ord);
    // the K2-gram order, singleton classes dropped (they have neither candidates nor walks)
    {uint32_t MM=WN-(K2-1); uint32_t*o=malloc(MM*4);for(uint32_t i=0;i<MM;i++)o[i]=i; KK=K2;qsort(o,MM,4,cmpk);
     uint32_t n=0;for(uint32_t j=0;j<MM;j++){int sp=j>0&&memcmp(W+o[j],W+o[j-1],K2)==0, sn=j+1<MM&&memcmp(W+o[j],W+o[j+1],K2)==0; if(sp||sn)o[n++]=o[j];}
     m2+=n; walk(o,n,K2,prev,K3?K3:258,&c2,&s2,&ch2,&r2,&a2); free(o);}
    if(K3){uint32_t MM=WN-(K3-1); uint32_t*o=malloc(MM*4);for(uint32_t i=0;i<MM;i++)o[i]=i; KK=K3;qsort(o,MM,4,cmpk);
     uint32_t n=0;for(uint32_t j=0;j<MM;j++){int sp=j>0&&memcmp(W+o[j],W+o[j-1],K3)==0, sn=j+1<MM&&memcmp(W+o[j],W+o[j+1],K3)==0; if(sp||sn)o[n++]=o[j];}
     m3+=n; walk(o,n,K3,prev,258,&c3,&s3,&ch3,&r3,&a3); free(o);}
  }
  printf("positions walked %lu; chunks of the 6-gram order with a walk %lu\n",npos,ch1);
  printf("one level  (6):        candidates/pos %6.2f  records/pos %.3f  walk steps per 6-gram chunk %6.2f  lanes alive per step %.1f\n",(double)c1/npos,(double)r1/npos,(double)s1/ch1,(double)a1/(s1?s1:1));
  printf("two levels (6,%d):      level 6: candidates/pos %6.2f records %.3f steps/chunk %.2f alive %.1f | level %d: order entries %.2f of the window, candidates/pos %6.2f records %.3f steps per 6-gram chunk %.2f (its own chunks: %lu) alive %.1f\n",
    K2,(double)c6/npos,(double)r6/npos,(double)s6/ch1,(double)a6/(s6?s6:1),K2,(double)m2/((double)npos*1.5),(double)c2/npos,(double)r2/npos,(double)s2/ch1,ch2,(double)a2/(s2?s2:1));
  if(K3) printf("third level (%d): order entries %.2f of the window, candidates/pos %6.2f records %.3f steps per 6-gram chunk %.2f (its own chunks: %lu)\n",K3,(double)m3/((double)npos*1.5),(double)c3/npos,(double)r3/npos,(double)s3/ch1,ch3);
  printf("TOTAL candidates/pos: one level %.2f, multi-level %.2f; walk steps per 6-gram chunk: one level %.2f, multi-level %.2f\n",(double)c1/npos,(double)(c6+c2+c3)/npos,(double)s1/ch1,(double)(s6+s2+s3)/ch1);
  return 0;}
