"""Where to put a large regressor output buffer (DESIGN.md section 3 "Output placement", profiles/r2/placement.txt).

On MI355X the rate of the regressor's multi-stream store pattern depends on the physical backing of the OUTPUT allocation: the same launch
runs at 0.45 ms into roughly one 2.88 GB allocation in ten and at 0.51-0.54 ms into the others, reproducibly per allocation.  A caller that
re-uses its output buffer picks a good one once: allocate candidates (earlier ones stay alive so that new ones land elsewhere), time the
launch it is going to repeat into each, keep the fastest, free the rest."""


def pick_output_buffer(launch, shape, device, max_candidates=72, batch=12, stand_out=0.86, mem_fraction=0.75, dtype=None):
    """launch(Y) runs the caller's kernel(s) into the candidate tensor Y (asynchronously, on the current stream).
    Returns (Y_best, info): info = {"candidates", "chosen", "probe_ms", "first_allocation_ms", "median_ms"}.
    Candidates are added `batch` at a time until one is `stand_out` x the median or better, `max_candidates` are reached or
    `mem_fraction` of the free device memory is in use."""
    import torch
    dtype = dtype or torch.float64
    nbytes = torch.empty((), dtype=dtype).element_size()
    for d in shape:
        nbytes *= d
    free_b, _ = torch.cuda.mem_get_info(device)
    max_c = max(1, min(int(max_candidates), int(mem_fraction * free_b) // max(1, nbytes)))

    def probe(Yc):
        launch(Yc)
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            launch(Yc)
        e1.record()
        torch.cuda.synchronize(device)
        return e0.elapsed_time(e1) / 3

    cands, probe_ms = [], []
    while len(cands) < max_c:
        for _ in range(min(batch, max_c - len(cands))):
            cands.append(torch.empty(shape, dtype=dtype, device=device))
            probe_ms.append(probe(cands[-1]))
        if len(cands) >= batch and min(probe_ms) <= stand_out * sorted(probe_ms)[len(probe_ms) // 2]:
            break
    best = min(range(len(cands)), key=lambda i: probe_ms[i])
    Y = cands[best]
    del cands
    torch.cuda.empty_cache()
    info = {"candidates": len(probe_ms), "chosen": best, "probe_ms": [round(t, 4) for t in probe_ms],
            "first_allocation_ms": round(probe_ms[0], 4), "median_ms": round(sorted(probe_ms)[len(probe_ms) // 2], 4)}
    return Y, info
