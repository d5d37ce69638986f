// AddressSanitizer / UBSan driver for the CPU oracle (oracle/vrc_oracle.c): terrain noise, SVO compile,
// frames (GI, 2 bounces, checkerboard blend, serial RNG), degenerate rays (zero / NaN / inf), grid DDA.
#include "vrc_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
int main(void){
  const int depth=7, S=128;
  int32_t* h=malloc(sizeof(int32_t)*S*S); orc_terrain_heights(S,1337,h);
  orc_svo* s=orc_svo_new(depth); orc_svo_fill_terrain(s,h); orc_lnode* nodes; uint64_t n=orc_svo_compile(s,&nodes);
  printf("nodes %lu\n",(unsigned long)n);
  uint8_t tex[768]; memset(tex,200,768);
  orc_camera cam={{64,8,64},{0},1.0f,0.3f,40.0f}; orc_make_rotation(0.2f,-0.5f,cam.rot);
  orc_frame_params p; memset(&p,0,sizeof(p)); p.width=96;p.height=56;p.light_position[0]=0.6f;p.light_position[1]=-0.95f;p.light_position[2]=0.41f;
  p.use_gi=1;p.use_samples=1;p.shadow_samples=0;p.gi_bounces=2;p.checker_parity=-1;p.seed=7;
  uint8_t* img=calloc(96*56*4,1); uint32_t* acc=calloc(96*56*4,4); orc_hit* prim=calloc(96*56,sizeof(orc_hit)); orc_frame_stats st;
  for(int f=0;f<3;f++){p.frame_index=f; orc_render_frame(nodes,depth,tex,tex,&cam,&p,img,acc,prim,&st,2);}
  p.use_samples=0;p.checker_parity=1; orc_render_frame(nodes,depth,tex,tex,&cam,&p,img,acc,prim,&st,1);
  p.rng_mode=1; orc_xorshf96_reset(); orc_render_frame(nodes,depth,tex,tex,&cam,&p,img,acc,prim,&st,1);
  printf("rays %lu steps %lu\n",(unsigned long)st.rays,(unsigned long)st.sum_complexity);
  float org[][3]={{1.5f,1.1f,1.5f},{0,0,0},{5,5,5},{1.5f,1.5f,1.5f},{NAN,1,1},{1.2f,1.3f,1.4f}};
  float dir[][3]={{0,1,0},{1,1,1},{-1,-1,-1},{0,0,0},{1,0,0},{INFINITY,1,0}};
  orc_hit out[6]; orc_cast_rays(nodes,depth,6,&org[0][0],&dir[0][0],NULL,NULL,out,1);
  for(int i=0;i<6;i++) printf("%u/%u ",out[i].hit,out[i].complexity); printf("\n");
  uint8_t cells[8*8*8]; memset(cells,0,512); cells[3*64+4*8+5]=1;
  float go[3]={0.5f,0.5f,0.5f},gd[3]={0.4f,0.5f,0.6f}; orc_hit gh; orc_grid_cast_ray(cells,8,8,8,go,gd,&gh); printf("grid %u %u\n",gh.hit,gh.complexity);
  /* round 2 surface: present chain (all medians, odd sizes), the reference-decomposition entries, the trace hooks */
  { const uint32_t W=37,H=23,OW=50,OH=31; uint8_t* im=malloc(W*H*4); uint8_t* den=calloc(W*H*4,1); uint8_t* win=malloc(OW*OH*4);
    for(uint32_t i=0;i<W*H*4;i++) im[i]=(uint8_t)(i*37u+11u);
    for(uint32_t m=0;m<3;m++){ orc_present(im,den,W,H,0.1f,m==0?0u:(m==1?3u:5u),win,OW,OH); orc_present(im,den,W,H,0.0f,0u,NULL,0,0); }
    printf("present %u %u\n",win[0],den[5]); free(im);free(den);free(win); }
  p.use_samples=1;p.checker_parity=-1;p.rng_mode=0;p.gi_bounces=1;p.width=99;p.height=57;
  { uint8_t* im=calloc(99*57*4,1); uint32_t* ac=calloc(99*57*4,4); orc_frame_stats s2;
    orc_render_frame_areas16(nodes,depth,tex,tex,&cam,&p,im,ac,&s2);
    orc_area_frame af; memset(&af,0,sizeof(af)); af.nodes=nodes;af.depth=depth;af.tex_top=tex;af.tex_side=tex;af.cam=&cam;af.params=&p;af.image=im;af.accum=ac;
    for(uint32_t t=0;t<17;t++) orc_area_job(t,16,&af);
    uint16_t* tr=calloc(99*57*ORC_TRACE_RAYS,2); uint8_t* ops=calloc((size_t)99*57*ORC_TRACE_RAYS*64,1);
    orc_render_frame_optrace(nodes,depth,tex,tex,&cam,&p,2,tr,ops,64);
    printf("areas %lu trace %u\n",(unsigned long)s2.rays,tr[0]); free(im);free(ac);free(tr);free(ops); }
  orc_free(nodes); orc_svo_free(s); free(h);free(img);free(acc);free(prim); return 0; }
