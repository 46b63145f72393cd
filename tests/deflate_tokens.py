"""Tiny RFC1951 tokenizer for debugging: returns [(out_pos, 'L', byte) | (out_pos, 'M', len, dist) | ('B', btype)]"""
LB=[3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LX=[0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DB=[1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DX=[0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]
def mk(lens):
    bl=[0]*16
    for l in lens: bl[l]+=1
    bl[0]=0; code=0; nxt=[0]*16
    for b in range(1,16):
        code=(code+bl[b-1])<<1; nxt[b]=code
    d={}
    for s,l in enumerate(lens):
        if l:
            d[(l,nxt[l])]=s; nxt[l]+=1
    return d
def tokens(data):
    pos=[0]
    def bit():
        b=(data[pos[0]>>3]>>(pos[0]&7))&1; pos[0]+=1; return b
    def bits(n):
        v=0
        for i in range(n): v|=bit()<<i
        return v
    def sym(d):
        c=0;l=0
        while True:
            c=(c<<1)|bit(); l+=1
            if (l,c) in d: return d[(l,c)]
    out=[]; op=0
    while True:
        last=bit(); t=bits(2); out.append(('B',t,op))
        if t==0:
            pos[0]=(pos[0]+7)&~7; n=bits(16); bits(16)
            for i in range(n): out.append((op,'L',bits(8))); op+=1
        else:
            if t==1:
                lt=mk([8]*144+[9]*112+[7]*24+[8]*8); dt=mk([5]*30)
            else:
                hl=bits(5)+257; hd=bits(5)+1; hc=bits(4)+4
                order=[16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15]
                cl=[0]*19
                for i in range(hc): cl[order[i]]=bits(3)
                ct=mk(cl); ls=[]
                while len(ls)<hl+hd:
                    s=sym(ct)
                    if s<16: ls.append(s)
                    elif s==16: ls+= [ls[-1]]*(3+bits(2))
                    elif s==17: ls+=[0]*(3+bits(3))
                    else: ls+=[0]*(11+bits(7))
                lt=mk(ls[:hl]); dt=mk(ls[hl:])
            while True:
                s=sym(lt)
                if s<256: out.append((op,'L',s)); op+=1
                elif s==256: break
                else:
                    l=LB[s-257]+bits(LX[s-257]); ds=sym(dt); d=DB[ds]+bits(DX[ds])
                    out.append((op,'M',l,d)); op+=l
        if last: break
    return out
