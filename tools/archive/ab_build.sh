# A/B of two builds of csrc/kvq_nn.hip on ONE box with a probe command
set -e
probe() { timeout -k 10 200 python tools/nn_probe.py 2>/dev/null | grep -E "attn|ln_" | tr '\n' ';'; echo; }
echo shipped; probe
touch kindergarten-vq-vae_amd/csrc/kvq_nn.hip
bash kindergarten-vq-vae_amd/build.sh "$@" > /dev/null
echo "variant($*)"; probe
