"""evc_set_pipeline(2): the two half batches of a step run on two side streams — which HIP does not promise to run concurrently.
The second engine of a process regularly got a pair whose kernels ran one after the other (its pipelined step then took 31 us
instead of 18); the engine now checks the pair with two spin kernels and replaces a second-half stream that does not overlap
the first (tools/probes/side_overlap.py)."""
import pytest

from helpers import make_workload

pytestmark = pytest.mark.gpu


def test_side_streams_of_every_engine_run_concurrently(caltech):
    import torch
    from sustaingym_amd.engine import StepEngine
    N = 4096
    wl = make_workload(caltech, N, bank_slots=16, seed=1)
    engines = []
    for i in range(4):                                   # all alive at once: the second one used to draw the bad pair
        e = StepEngine(caltech, N, project_action=True, autoreset=True, bank_slots=16, max_sessions=wl['sessions'].shape[1],
                       moer_days=wl['moer'].shape[0])
        e.upload_moer(wl['moer'])
        e.upload_episodes(wl['n_sessions'], wl['sessions'], wl['requested'], wl['moer_day'])
        e.reset()
        e.set_pipeline(2)
        engines.append(e)

    def spin(streams):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for st in streams:
            st.wait_event(a)
            with torch.cuda.stream(st):
                torch.cuda._sleep(400000)
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b)

    for i, e in enumerate(engines):
        streams = [st for _, st in e.pipeline_halves()]
        one = min(spin(streams[:1]) for _ in range(3))
        both = min(spin(streams) for _ in range(3))
        assert both < 1.5 * one, (i, one, both)          # serialised: 1.9; concurrent: 1.03 - 1.08
    for e in engines:
        e.close()
