import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import femus_amd
from femus_amd.poisson import PoissonMG
ctx = femus_amd.Context(0)
pb = PoissonMG(ctx, 8, 8, 8, 4).init()
pb.assemble(); pb.prepare(); pb.assemble(); pb.prepare()
ctx.sync()
pb.assemble()
ctx.set_option("asm_debug", 128)
pb.prepare()
ctx.set_option("asm_debug", 0)
