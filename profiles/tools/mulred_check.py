import random
P=0xffffffff00000001; EPS=0xffffffff; M32=(1<<32)-1; M64=(1<<64)-1
def mulred(a,b):
    a0,a1,b0,b1=a&M32,a>>32,b&M32,b>>32
    Pq=a0*b0                                  # mad
    M=a0*b1
    Q=a1*b1
    M=M+a1*b0; cm=M>>64; M&=M64              # mad with carry
    P0,P1=Pq&M32,Pq>>32
    t=P1+(M&M32); c1=t>>32; P1=t&M32         # add_co
    t=(M>>32)+(Q&M32)+c1; c2=t>>32; q=t&M32  # addc_co
    T=(P0|(P1<<32))+q*EPS; c3=T>>64; T&=M64  # mad carry
    t=(Q>>32)+c2; assert t<=M32; h=t         # addc (carry ignored)
    e=c3
    t=h+cm; assert t<=M32, (hex(a),hex(b)); h=t
    T0,T1=T&M32,T>>32
    t=T0-h; br=1 if t<0 else 0; U0=t&M32
    t=T1-br; b_=1 if t<0 else 0; U1=t&M32
    e=e-b_                                   # signed
    U=U0|(U1<<32)
    V=(U-e)&M64                              # mad_i64_i32(e,-1,U)
    r1=((V>>32)+e)&M32
    r=(V&M32)|(r1<<32)
    return r
random.seed(1)
edge=[0,1,2,EPS,EPS+1,EPS-1,P-1,P,P+1,M64,M64-1,1<<63,(1<<63)-1,0xffffffff00000000,0xfffffffeffffffff,0x00000001ffffffff,0xffffffff,0x100000000]
vals=edge+[random.getrandbits(64) for _ in range(300)]+[random.getrandbits(32)<<32 for _ in range(20)]+[(random.getrandbits(32)<<32)|M32 for _ in range(20)]+[((M32)<<32)|random.getrandbits(32) for _ in range(20)]
n=0
for a in vals:
    for b in vals:
        r=mulred(a,b)
        assert 0<=r<=M64
        assert r%P==(a*b)%P,(hex(a),hex(b),hex(r))
        n+=1
print('ok',n)
