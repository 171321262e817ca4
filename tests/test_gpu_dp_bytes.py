"""-m gpu, single process: wire bytes of the three camera-DP exchanges on the BENCH scenes, by partitioning
them into 2 / 4 / 8 VIRTUAL ranks (clm_gs_amd.dp.exchange_bytes: pure index arithmetic on the touched sets
the real visibility pass selects; the same model is checked against the counters of the real collectives in
tests/test_dp_gloo.py).  No multi-GPU hardware is involved and no scaling figure is claimed; the numbers say
how many bytes each rank must SEND per batch:

  allreduce  240 B x union of touched rows, all-reduced (+ the touched mask)          [round-2 default]
  owner      all-gather of parameter rows / reduce-scatter of gradient rows, padded     [dp_owner_computes]
  locality   only border rows travel + owners publish 52 B per touched row per peer     [dp_locality]

each with the round-2 camera order (rank r takes cameras r::G of a shuffled global batch) and with the locality
deal (dp.deal_cameras).  Report: gpurun_out/dp_bytes.json (committed copy under profiles/)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [("rubble28m", 28_000_000, 4608, 3456, 4, 0.10), ("bigcity102m", 102_231_360, 1920, 1080, 8, 0.02)]


@pytest.mark.parametrize("name,N,W,H,bsz,vis", CASES)
def test_exchange_bytes_of_virtual_ranks(dev, name, N, W, H, bsz, vis):
    from types import SimpleNamespace

    from clm_gs_amd import dp, utils
    from clm_gs_amd.strategies.base_engine import select_filters
    from clm_gs_amd.synthetic import nadir_cameras, synth_gaussians
    args = utils.default_args(bsz=bsz)
    args.clm_offload = True
    utils.set_args(args)
    utils.set_img_size(H, W)
    sc = synth_gaussians(N, seed=0, device="cuda")
    order = utils.morton_order(sc["xyz"])
    g = SimpleNamespace(_xyz=utils.gather_rows(sc["xyz"], order), _scaling=utils.gather_rows(sc["scaling"], order),
                        _rotation=utils.gather_rows(sc["rotation"], order))
    del sc, order
    steps = 3
    report = {"config": name, "n_gaussians": N, "bsz_per_rank": bsz, "steps_averaged": steps,
              "camera_population_per_rank": 25 * bsz, "ranks": {}}
    for G in (2, 4, 8):
        pop = 25  # the camera population the deal chooses from: the bench's 25 batches per rank
        cams = nadir_cameras(pop * bsz * G, N, W, H, vis, seed=0, device="cuda")
        perm = torch.randperm(len(cams), generator=torch.Generator().manual_seed(7)).tolist()
        cams = [cams[i] for i in perm]                       # the bench's shuffled order
        ranks_of, shares = dp.deal_cameras(cams, g, G)
        pools = [[c for c, q in zip(cams, ranks_of) if q == r] for r in range(G)]
        assert all(len(p) == pop * bsz for p in pools)
        local_share = sum(int(shares[c, q]) for c, q in enumerate(ranks_of)) / float(shares.sum())
        acc = {}
        for deal_name in ("strided", "locality_deal"):
            tot = {"allreduce": 0.0, "owner": 0.0, "locality": 0.0, "locality_small_owner": 0.0, "union": 0.0,
                   "ref240": 0.0, "border": 0.0, "touched": 0.0}
            for s in range(steps):
                T = []
                for r in range(G):
                    if deal_name == "strided":
                        batch = cams[s * bsz * G:(s + 1) * bsz * G][r::G]
                    else:
                        batch = pools[r][s * bsz:(s + 1) * bsz]
                    with torch.no_grad():
                        _, tr = select_filters(batch, g._xyz, g._scaling, g._rotation)
                    T.append(tr.long())
                b = dp.exchange_bytes(T, N)
                for k in ("allreduce", "owner", "locality", "locality_small_owner"):
                    tot[k] += max(b[k]) / steps           # the slowest rank sets the pace
                tot["union"] += b["union"] / steps
                tot["ref240"] += b["reference_240B_x_union"] / steps
                tot["border"] += max(b["border"]) / steps
                tot["touched"] += max(b["touched"]) / steps
                del T
            acc[deal_name] = {k: round(v, 1) for k, v in tot.items()}
        # round 4: the small attributes at their owners (no step F) never cost more than the published sums did
        assert acc["locality_deal"]["locality_small_owner"] <= acc["locality_deal"]["locality"] * 1.02
        best = acc["locality_deal"]["locality"]
        report["ranks"][str(G)] = dict(
            acc, local_share_of_the_deal=round(local_share, 4),
            locality_vs_240B_union=round(acc["locality_deal"]["ref240"] / best, 2),
            locality_vs_round2_allreduce=round(acc["strided"]["allreduce"] / best, 2))
        # the locality exchange on the locality deal sends less than every alternative
        assert best < acc["strided"]["allreduce"] and best < acc["locality_deal"]["allreduce"]
        assert best < acc["locality_deal"]["owner"]
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "dp_bytes.json")
    allr = json.load(open(path)) if os.path.exists(path) else {}
    allr[name] = report
    json.dump(allr, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(report))
