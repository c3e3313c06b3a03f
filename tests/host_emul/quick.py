import ctypes, os, numpy as np, time
R = ctypes.CDLL("oracle/_ref/libsecp256k1_ref.so")
E = ctypes.CDLL("tests/host_emul/libs2k_hostemu.so")
rng = np.random.default_rng(7)
P = 2**256 - 2**32 - 977
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
def rb(n=32): return bytes(rng.integers(0,256,n,dtype=np.uint8))
def b(v): return int(v).to_bytes(32,'big')
edge = [0,1,2,P-1,P-2,P,P+1,2**256-1,2**255,977,2**32+977, N, N-1, N+1, (P+1)//2]
def fe_cases(k):
    out=[b(e % 2**256) for e in edge]
    out += [rb() for _ in range(k)]
    return out
def call(lib,name,nout,*args):
    outs=[ctypes.create_string_buffer(n) for n in nout]
    r=getattr(lib,name)(*outs,*args)
    return r,[o.raw for o in outs]
bad=0
cs=fe_cases(300)
for i in range(len(cs)):
    a=cs[i]; c=cs[(i*7+3)%len(cs)]; d=cs[(i*11+5)%len(cs)]
    for name,args in (("fe_mul",(a,c)),("fe_sqr",(a,)),("fe_add",(a,c)),("fe_negate",(a,)),("fe_inv",(a,)),("fe_sqrt",(a,))):
        r1,o1=call(R,"ref_"+name,[32],*args); r2,o2=call(E,"emu_"+name,[32],*args)
        if o1!=o2 or (name=="fe_sqrt" and r1!=r2): bad+=1; print("MISMATCH",name,a.hex(),c.hex(),o1[0].hex(),o2[0].hex(),r1,r2)
    for name,args in (("scalar_mul",(a,c)),("scalar_add",(a,c)),("scalar_negate",(a,)),("scalar_set_b32",(a,))):
        r1,o1=call(R,"ref_"+name,[32],*args); r2,o2=call(E,"emu_"+name,[32],*args)
        if o1!=o2 or (name=="scalar_set_b32" and r1!=r2): bad+=1; print("MISMATCH",name,a.hex(),c.hex())
    r1,o1=call(R,"ref_scalar_split_lambda",[32,32],a); r2,o2=call(E,"emu_scalar_split_lambda",[32,32],a)
    if o1!=o2: bad+=1; print("MISMATCH split",a.hex())
    if i<40:
        r1,o1=call(R,"ref_scalar_inverse",[32],a); r2,o2=call(E,"emu_scalar_inverse",[32],a)
        if o1!=o2: bad+=1; print("MISMATCH scinv",a.hex())
print("field/scalar mismatches:",bad)
# points
def rand_point():
    while True:
        x=rb(); r,o=call(R,"ref_ge_set_xquad",[64],x)
        if r: return o[0]
pts=[rand_point() for _ in range(20)]
G=bytes.fromhex("79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8")
pts.append(G)
def neg(p): return p[:32]+b((P-int.from_bytes(p[32:],'big'))%P)
bad=0
for i,a in enumerate(pts):
    for c in (pts[(i+1)%len(pts)], a, neg(a)):
        for ai in (0,1):
            for bi in (0,1):
                r1,o1=call(R,"ref_ge_add",[64],a,ai,c,bi); r2,o2=call(E,"emu_ge_add",[64],a,ai,c,bi)
                if (r1,o1)!=(r2,o2): bad+=1; print("MISMATCH ge_add",i,ai,bi)
                za=rb(); zb=rb()
                r2,o2=call(E,"emu_gej_add_var",[64],a,ai,c,bi,za,zb)
                if (r1,o1)!=(r2,o2): bad+=1; print("MISMATCH gej_add_var",i,ai,bi)
    r1,o1=call(R,"ref_ge_double",[64],a,0); r2,o2=call(E,"emu_ge_double",[64],a,0)
    if (r1,o1)!=(r2,o2): bad+=1; print("MISMATCH dbl")
print("group mismatches:",bad)
# ecmult
t=time.time(); bad=0
sc_edge=[b(0),b(1),b(2),b(N-1),b(N-2),b(3),b(255),b(256),b(2**128),b(2**128-1),b(N//2),b(N//2+1)]
cases=[]
for i in range(60):
    a=pts[i%len(pts)]; na=rb() if i%3 else sc_edge[i%len(sc_edge)]; ng=rb() if i%4 else sc_edge[(i*5)%len(sc_edge)]
    cases.append((a,0,na,ng))
cases += [(G,0,b(1),b(1)),(G,0,b(1),b(N-1)),(G,0,b(2),b(N-2)),(G,1,b(5),b(7)),(G,0,b(0),b(0)),(G,0,b(5),None),(pts[0],0,rb(),None),(G,0,b(3),b(3)),(neg(G),0,b(1),b(2)), (G,0,b(255),b(1))]
for (a,ai,na,ng) in cases:
    r1,o1=call(R,"ref_ecmult",[64],a,ai,na,ng); 
    for z in (None, rb()):
        r2,o2=call(E,"emu_ecmult",[64],a,ai,na,ng,z)
        if (r1,o1)!=(r2,o2): bad+=1; print("MISMATCH ecmult",a.hex()[:8],ai,na.hex(),ng.hex() if ng else None, r1,r2)
print("ecmult mismatches:",bad,"of",len(cases)*2, "time",time.time()-t)
import hashlib
for n in (0,1,55,56,63,64,65,119,120,1000):
    m=rb(n) if n else b''; r,o=call(E,"emu_sha256",[32],m,ctypes.c_size_t(n))
    assert o[0]==hashlib.sha256(m).digest(), n
print("sha ok")
