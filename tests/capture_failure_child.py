"""Child process of tests/test_discriminator_gpu.py::test_train_engine_capture_failure_drops_every_graph (GPU only).
usage: capture_failure_child.py g|d   - which half's hipGraph capture is sabotaged."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "srgan-st_amd")]
import torch  # noqa: E402


def make_cfg(ch, rcb, dch):
    from srganst.config import Config
    cfg = Config()
    cfg.MODEL.G_N_CHANNEL, cfg.MODEL.G_N_RCB, cfg.MODEL.D_N_CHANNEL = ch, rcb, dch
    return cfg


INTERVAL = 2      # D updated every second iteration: the engine holds TWO graphs (merged G + D iteration, generator-only iteration)


def run(use_graph, sabotage):
    from srganst.engine import TrainEngine
    from srganst.loss import MSELoss, StructureTensorLoss
    from srganst.model import Discriminator, Generator
    cfg = make_cfg(16, 2, 8)
    torch.manual_seed(1)
    D, G = Discriminator(cfg).cuda().train(), Generator(cfg).cuda().train()
    cfg.add_g_criterion("Pixel", MSELoss(), 1.0)
    cfg.add_g_criterion("ST", StructureTensorLoss(), 1 / 3)
    cfg.SOLVER.D_UPDATE_INTERVAL = INTERVAL
    eng = TrainEngine(cfg, G, D, use_graph=use_graph, adam_capturable=True)
    assert len(eng._steps()) == 2
    if sabotage:
        step = eng._g_fb if sabotage == "g" else eng._it       # "g": the generator-only graph, "d": the merged G + D iteration
        inner = step.fn

        def fn():
            if torch.cuda.is_current_stream_capturing():
                torch.cuda.synchronize()           # not allowed under capture: the capture fails
            return inner()
        step.fn = fn
    gen = torch.Generator().manual_seed(2)
    for _ in range(8):
        eng.step(torch.rand(4, 3, 96, 96, generator=gen).cuda(), torch.rand(4, 3, 24, 24, generator=gen).cuda())
    torch.cuda.synchronize()
    return eng, G.state_dict(), D.state_dict(), {k: v.item() for k, v in eng.loss_values.items()}


def main():
    fail = sys.argv[1]
    e0 = run(True, None)[0]
    assert e0.graph_active                                   # captures work in this process before the sabotage
    e1, g1, d1, l1 = run(False, None)
    e2, g2, d2, l2 = run(True, fail)
    assert not e1.graph_active and not e2.graph_active
    assert all(s.graph is None and not s.enabled for s in e2._steps())
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
    for k in d1:
        assert torch.equal(d1[k], d2[k]), k
    assert l1 == l2
    print("FALLBACK-PARITY-OK", flush=True)
    e3, g3, d3, l3 = run(True, None)                         # a later engine of the same process: captures again, same results
    assert e3.graph_active
    for k in g1:
        assert torch.equal(g1[k], g3[k]), k
    for k in d1:
        assert torch.equal(d1[k], d3[k]), k
    print("RECAPTURE-OK", flush=True)


if __name__ == "__main__":
    main()
