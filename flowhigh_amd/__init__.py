"""flowhigh_amd -- MI355X-native FlowHighSR.generate().

    from flowhigh_amd import FlowHighSR
    model = FlowHighSR.from_local(ckpt_dir, device="cuda")      # or .from_pretrained(device)
    wav48 = model.generate(audio, sr_in, 48000, timestep=1)     # torch [1, T48] on the GPU
"""
from .flowhighsr import FLowHigh, FlowHighSR  # noqa: F401
