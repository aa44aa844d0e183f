import sys, json; sys.path.insert(0, '/root/repo')
import torch, bench
import vllm_rust_amd as vmx
vmx.init(0)
import os
ctxs = tuple(int(c) for c in os.environ.get('CTXS', '1024,8192').split(','))
print(json.dumps(bench.context_sweep_secondary(32, 4, ctxs)['by_context']))
