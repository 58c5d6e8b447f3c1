import torch, sys
sys.path.insert(0, '/root/repo')
from foundpose_amd import ops
torch.manual_seed(0)
dev='cuda'
M, K = 1280*3, 1024
for n, epi in ((3072, 0), (4096, 1)):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, K, device=dev) * 0.02).to(torch.bfloat16)
    bias = torch.randn(n, device=dev)
    o1 = ops.gemm_bf16(a, w, bias, epilogue=epi | (256 << 8), m_valid=M - 700)
    o2 = ops.gemm_bf16(a, w, bias, epilogue=epi | (320 << 8), m_valid=M - 700)
    print(n, epi, torch.equal(o1[:M-700], o2[:M-700]), float((o1[:M-700].float()-o2[:M-700].float()).abs().max()), float(o2[M-700:].abs().max()))
M, N = 1280*3, 1024
for K in (1024, 4096):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    xb0 = torch.randn(M, N, device=dev).to(torch.bfloat16); xl0 = (torch.randn(M, N, device=dev)*0.003).to(torch.bfloat16)
    outs = []
    for tile in (256, 320):
        xb, xl = xb0.clone(), xl0.clone()
        st = ops.gemm_bf16_resid_hilo(a, w, bias, xb, xl, tile=tile, m_valid=M - 700)
        outs.append((xb, xl, st))
    v = M - 700
    print('hilo', K, torch.equal(outs[0][0][:v], outs[1][0][:v]), torch.equal(outs[0][1][:v], outs[1][1][:v]), torch.equal(outs[0][2][:, :v], outs[1][2][:, :v]),
          torch.equal(outs[1][0][v:], xb0[v:]), torch.equal(outs[1][1][v:], xl0[v:]))
