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
  orc_free(nodes); orc_svo_free(s); free(h);free(img);free(acc);free(prim); return 0; }
