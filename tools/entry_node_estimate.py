#!/usr/bin/env python3
"""tools/entry_node_estimate.py — what per-tile entry nodes for primary rays could save, priced on the real tree and camera
(large scene, 1200x800): for a third of the tiles, seven primary rays each are walked through the tree (geometric slab test, boxes + 0.02); the tile's
entry node is the deepest node below which all of them stay; printed: node visits per primary ray and how many of them lie above the entry node."""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
import rays1bench_amd as r1
from rays1bench_amd import binding
w,h=1200,800
sc=r1.create_large_scene(w,h)
info,nodes,ids=binding.bvh_describe(sc.spheres.contents if hasattr(sc.spheres,'contents') else sc.spheres)
print(info)
cam=sc.camera.contents
org=np.array(list(cam.origin)); ll=np.array(list(cam.lower_left)); hor=np.array(list(cam.horizontal)); ver=np.array(list(cam.vertical))
N=nodes.shape[0]
# node layout: {m0x m1x m0y m1y} {m0z m1z e0x e1x} {e0y e1y e0z e1z} {A K child0 child1}
m=np.stack([nodes[:,[0,2,4]],nodes[:,[1,3,5]]],1)  # [N,2,3]
e=np.stack([nodes[:,[6,8,10]],nodes[:,[7,9,11]]],1)
child=nodes[:,14:16].view(np.uint32)
LEAF=0x80000000
parent=np.full(N,-1); depth=np.zeros(N,int)
for i in range(N):
    for c in child[i]:
        if not (c & LEAF): parent[c]=i
order=[0]
for i in order:
    for c in child[i]:
        if not (c&LEAF): depth[c]=depth[i]+1; order.append(int(c))
def walk(o,d):
    inv=1.0/np.where(d==0,1e-30,d)
    visited=[]
    stack=[0]
    while stack:
        n=stack.pop(); visited.append(n)
        for k in range(2):
            a=(m[n,k]-o)*inv; b=(e[n,k]+0.02)*np.abs(inv)
            tn=np.max(a-b); tf=np.min(a+b)
            if tn<=tf and tf>=0:
                c=child[n,k]
                if not (c&LEAF): stack.append(int(c))
    return visited
def lca(a,b):
    while a!=b:
        if depth[a]>=depth[b]: a=parent[a]
        else: b=parent[b]
    return a
tx,ty=(w+31)//32,(h+31)//32
saved=[];vis=[]
rng=np.random.default_rng(1)
for t in range(0,tx*ty,3):
    x0,y0=(t%tx)*32,(t//tx)*32
    sets=[]
    for (fx,fy) in [(0,0),(1,0),(0,1),(1,1),(.5,.5),(.25,.75),(.75,.25)]:
        s_=(x0+32*fx)/w; t_=(y0+32*fy)/h
        d=ll+s_*hor+t_*ver-org; d/=np.linalg.norm(d)
        sets.append(walk(org,d))
    allv=set(n for s in sets for n in s)
    # walk down from the root while exactly one inner child was visited by any ray of the tile (and the other child is not a visited leaf... leaves ignored here)
    x=0; chain=[0]
    while True:
        kids=[int(c) for c in child[x] if not (c&LEAF) and int(c) in allv]
        # if a LEAF child of x could be hit we cannot skip x: approximate by requiring both children inner or the leaf child's box missed by all rays
        leafkid=[k for k in range(2) if (child[x,k]&LEAF)]
        hit_leaf=False
        for k in leafkid:
            for (fx,fy) in [(0,0),(1,0),(0,1),(1,1),(.5,.5)]:
                s_=(x0+32*fx)/w; t_=(y0+32*fy)/h
                d=ll+s_*hor+t_*ver-org; d/=np.linalg.norm(d)
                inv=1.0/np.where(d==0,1e-30,d)
                a=(m[x,k]-org)*inv; b=(e[x,k]+0.02)*np.abs(inv)
                if np.max(a-b)<=np.min(a+b) and np.min(a+b)>=0: hit_leaf=True
        if len(kids)==1 and not hit_leaf and x!=0:
            x=kids[0]; chain.append(x)
        elif x==0:
            x=1; chain.append(1)   # root step handles node 0
        else: break
    for s in sets:
        above=[n for n in s if n in chain[1:-1]]
        saved.append(len(above)); vis.append(len(s))
print('mean visits per primary ray (incl root)',np.mean(vis),'mean saved by starting at the tile LCA',np.mean(saved))
