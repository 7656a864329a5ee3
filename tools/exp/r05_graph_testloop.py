"""r05: the tests of tests/test_gpu_graph.py called in file order, over and over, in ONE process (the hunt for the one segmentation fault inside hipGraphLaunch that a
full `pytest -m gpu` session produced in test_sibling_steps_replay_bitwise).   python tools/exp/r05_graph_testloop.py [repetitions] [only-siblings]"""
import faulthandler, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
faulthandler.enable()
import torch
import conftest  # noqa: F401
import test_gpu_graph as T
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
only = len(sys.argv) > 2
calls = []
if not only:
    calls += [(T.test_replayed_steps_are_bitwise_the_eager_steps, a) for a in ("simclr", "byol", "barlow")]
    calls += [(T.test_graph_kernel_selection_trains_like_the_eager_one, None), (T.test_adamw_with_the_step_count_in_device_memory_is_the_by_value_update, None),
              (T.test_dino_step_replays_as_a_graph_through_an_epoch_schedule_change, None)]
calls += [(T.test_sibling_steps_replay_bitwise, a) for a in ("simsiam", "relic", "moco")]
if not only:
    calls += [(T.test_graph_survives_eager_work_between_replays_over_many_steps, None)]
t0 = time.time()
for r in range(reps):
    for fn, arg in calls:
        print(r, fn.__name__, arg, round(time.time() - t0, 1), flush=True)
        fn(dev, arg) if arg is not None else fn(dev)
print("done", reps, round(time.time() - t0, 1), flush=True)
