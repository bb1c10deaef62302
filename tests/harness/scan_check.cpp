// CPU check of csrc/nmpc_scan.h (tests/test_scan_riccati.py): stage elements + the combine rule in float32, as a Hillis-Steele suffix scan over
// the stages, against the sequential Riccati recursion in float64 on 2000 random stage problems with 30 % of the inputs held.
#include <cstdio>
#include <cmath>
#include <random>
#include "nmpc_scan.h"
using namespace nmpc;
int main(){
  std::mt19937 rng(1); std::normal_distribution<double> nd(0,1); std::uniform_real_distribution<double> ud(0,1);
  double worst=0;
  for(int trial=0;trial<2000;++trial){
    const int N=20;
    struct St{double a,b,B00,B01,B10,B11,B20,d[3],Q[6],q[3],R00,R01,R11,r[2];int st0,st1;double v0,v1;} st[N];
    for(int k=0;k<N;++k){ auto&s=st[k]; s.a=0.01*nd(rng); s.b=0.01*nd(rng); s.B00=0.01*nd(rng);s.B01=0.01*nd(rng);s.B10=0.01*nd(rng);s.B11=0.01*nd(rng);s.B20=0.03*nd(rng);
      for(int i=0;i<3;++i){s.d[i]=1e-3*nd(rng); s.q[i]=nd(rng);} double g[9]; for(auto&x:g)x=nd(rng);
      // Q = G G' + diag
      double Qm[3][3]; for(int i=0;i<3;++i)for(int j=0;j<3;++j){Qm[i][j]=(i==j?5.0:0.0); for(int l=0;l<3;++l)Qm[i][j]+=g[3*i+l]*g[3*j+l];}
      s.Q[0]=Qm[0][0];s.Q[1]=Qm[0][1];s.Q[2]=Qm[0][2];s.Q[3]=Qm[1][1];s.Q[4]=Qm[1][2];s.Q[5]=Qm[2][2];
      s.R00=0.1+ud(rng); s.R11=0.1+ud(rng); s.R01=0.05*nd(rng); s.r[0]=nd(rng); s.r[1]=nd(rng);
      s.st0 = ud(rng)<0.3 ? (ud(rng)<0.5?ST_LOWER:ST_UPPER) : ST_FREE; s.st1 = ud(rng)<0.3 ? ST_LOWER : ST_FREE; s.v0=0.5*nd(rng); s.v1=0.5*nd(rng);}
    double PN[6]={10,0.5,0.2,8,0.1,2}, pN[3]={nd(rng),nd(rng),nd(rng)};
    // double sequential: P_k, p_k
    double P[3][3]={{PN[0],PN[1],PN[2]},{PN[1],PN[3],PN[4]},{PN[2],PN[4],PN[5]}}, p[3]={pN[0],pN[1],pN[2]};
    double Pk[N+1][6], pk[N+1][3];
    auto save=[&](int k){Pk[k][0]=P[0][0];Pk[k][1]=P[0][1];Pk[k][2]=P[0][2];Pk[k][3]=P[1][1];Pk[k][4]=P[1][2];Pk[k][5]=P[2][2];for(int i=0;i<3;++i)pk[k][i]=p[i];};
    save(N);
    for(int k=N-1;k>=0;--k){ auto&s=st[k]; double A[3][3]={{1,0,s.a},{0,1,s.b},{0,0,1}}, B[3][2]={{s.B00,s.B01},{s.B10,s.B11},{s.B20,-s.B20}};
      double Q[3][3]={{s.Q[0],s.Q[1],s.Q[2]},{s.Q[1],s.Q[3],s.Q[4]},{s.Q[2],s.Q[4],s.Q[5]}}; double R[2][2]={{s.R00,s.R01},{s.R01,s.R11}};
      bool f[2]={s.st0==ST_FREE,s.st1==ST_FREE}; double hv[2]={f[0]?0:s.v0,f[1]?0:s.v1};
      double dd[3],rr[2]; for(int i=0;i<3;++i)dd[i]=s.d[i]+B[i][0]*hv[0]+B[i][1]*hv[1]; rr[0]=s.r[0]+R[0][1]*hv[1]; rr[1]=s.r[1]+R[0][1]*hv[0];
      double sv[3]; for(int i=0;i<3;++i){sv[i]=p[i];for(int j=0;j<3;++j)sv[i]+=P[i][j]*dd[j];}
      double PB[3][2],PA[3][3]; for(int i=0;i<3;++i){for(int c=0;c<2;++c){PB[i][c]=0;for(int j=0;j<3;++j)PB[i][c]+=P[i][j]*B[j][c];} for(int c=0;c<3;++c){PA[i][c]=0;for(int j=0;j<3;++j)PA[i][c]+=P[i][j]*A[j][c];}}
      double H[2][2],G[2][3],hu[2]; for(int a=0;a<2;++a){for(int c=0;c<2;++c){H[a][c]=R[a][c];for(int i=0;i<3;++i)H[a][c]+=B[i][a]*PB[i][c];} for(int c=0;c<3;++c){G[a][c]=0;for(int i=0;i<3;++i)G[a][c]+=B[i][a]*PA[i][c];} hu[a]=rr[a];for(int i=0;i<3;++i)hu[a]+=B[i][a]*sv[i];}
      // masked inverse over free inputs
      double Hi[2][2]={{0,0},{0,0}}; if(f[0]&&f[1]){double det=H[0][0]*H[1][1]-H[0][1]*H[0][1];Hi[0][0]=H[1][1]/det;Hi[1][1]=H[0][0]/det;Hi[0][1]=Hi[1][0]=-H[0][1]/det;} else if(f[0])Hi[0][0]=1/H[0][0]; else if(f[1])Hi[1][1]=1/H[1][1];
      double Pn[3][3],pn[3];
      for(int i=0;i<3;++i){ for(int j=0;j<3;++j){ double v=Q[i][j]; for(int l=0;l<3;++l)v+=A[l][i]*PA[l][j]; for(int a=0;a<2;++a)for(int c=0;c<2;++c)v-=G[a][i]*Hi[a][c]*G[c][j]; Pn[i][j]=v;}
        double v=s.q[i]; for(int l=0;l<3;++l)v+=A[l][i]*sv[l]; for(int a=0;a<2;++a)for(int c=0;c<2;++c)v-=G[a][i]*Hi[a][c]*hu[c]; pn[i]=v;}
      for(int i=0;i<3;++i){for(int j=0;j<3;++j)P[i][j]=0.5*(Pn[i][j]+Pn[j][i]);p[i]=pn[i];}
      save(k);}
    // float scan: elements, Hillis-Steele suffix scan over N+1 positions
    ScanElem el[N+1]; bool bad;
    for(int k=0;k<N;++k){auto&s=st[k]; scan_stage_element((float)s.a,(float)s.b,(float)s.B00,(float)s.B01,(float)s.B10,(float)s.B11,(float)s.B20,(float)s.d[0],(float)s.d[1],(float)s.d[2],(float)s.Q[0],(float)s.Q[1],(float)s.Q[2],(float)s.Q[3],(float)s.Q[4],(float)s.Q[5],(float)s.q[0],(float)s.q[1],(float)s.q[2],(float)s.R00,(float)s.R01,(float)s.R11,(float)s.r[0],(float)s.r[1],s.st0,s.st1,(float)s.v0,(float)s.v1,el[k],bad); if(bad){printf("bad\n");return 1;}}
    scan_identity(el[N]); for(int i=0;i<9;++i)el[N].A[i]=(i%4==0)?1.f:0.f; for(int i=0;i<6;++i)el[N].J[i]=(float)PN[i]; for(int i=0;i<3;++i)el[N].h[i]=(float)pN[i];
    for(int s=1;s<N+1;s*=2){ ScanElem nw[N+1]; for(int k=0;k<=N;++k){ if(k+s<=N) scan_combine(el[k],el[k+s],nw[k]); else nw[k]=el[k]; } for(int k=0;k<=N;++k)el[k]=nw[k]; }
    for(int k=0;k<=N;++k){ double scale=0; for(int i=0;i<6;++i)scale=std::fmax(scale,std::fabs(Pk[k][i])); double ps=1e-9; for(int i=0;i<3;++i)ps=std::fmax(ps,std::fabs(pk[k][i]));
      for(int i=0;i<6;++i)worst=std::fmax(worst,std::fabs(el[k].J[i]-Pk[k][i])/scale); for(int i=0;i<3;++i)worst=std::fmax(worst,std::fabs(el[k].h[i]-pk[k][i])/ps);}
  }
  printf("worst relative deviation of (P_k, p_k) scan float32 vs sequential float64: %.3e\n",worst);
  return worst<2e-4?0:1;
}
