"""Worker of tests/test_gpu_product_library.py::test_a_call_under_stream_capture_does_not_calibrate (GPU box only)."""
import os
import sys

import numpy as np
import torch  # first: its HIP runtime must be the one the process initialises (streams and graphs are shared with the library)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stormphrax_amd as sp  # noqa: E402


def main():
    pos = sp.random_positions(20000, seed=77)
    blob = sp.synthetic_net_bytes("tame")
    d_pos = torch.from_numpy(pos.view(np.uint8).reshape(-1, 32).copy()).cuda()
    d_out = torch.full((len(pos),), -1, dtype=torch.int32, device="cuda")
    with sp.NnueState(sp.Network(blob), device=0, max_batch=len(pos), sliced_ft=False) as plain:
        want = plain.evaluate_once(pos)
    with sp.NnueState(sp.Network(blob), device=0, max_batch=len(pos), options={"ftx_auto_calibrate": 0}) as st:
        assert st.takes_sliced_pipeline(len(pos))
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            # the pipeline's tables and scratch are allocated by the first call (no allocation may happen inside a capture)
            st.evaluate_once_device(d_pos.data_ptr(), len(pos), d_out.data_ptr(), side.cuda_stream)
        side.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), want) and st.hot_rows().size == 0
        st.set_option("ftx_auto_calibrate", 1)
        d_out.fill_(-1)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side, capture_error_mode="relaxed"):
            st.evaluate_once_device(d_pos.data_ptr(), len(pos), d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert st.hot_rows().size == 0  # (captured: nothing was calibrated, nothing ran)
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), want)
        d_out.fill_(-1)
        st.evaluate_once_device(d_pos.data_ptr(), len(pos), d_out.data_ptr())
        st.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), want) and st.hot_rows().size > 0
        graph.replay()  # (the graph still holds the kernels of the empty set: same scores)
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), want)
    print("capture ok")


if __name__ == "__main__":
    main()
