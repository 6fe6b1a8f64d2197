class Config(object):
    """Same attribute names and defaults as the reference's config.py:1-20 (drop-in), plus a
    few optional attributes of the MI355X build that default to the reference's behaviour."""
    epoch = 5000
    batch_size = 8
    learning_rate = 0.002

    sigma = 1.0

    cuda = True
    gpu_cnt = 4               # reference: nn.DataParallel device count.  Here: informational; the
                              # number of ranks comes from torch.distributed (one process per GPU)

    async_loading = True
    pin_memory = True

    root_path = '/home/D2019063/MusicGeneration_VAE-torch'
    data_path = 'data/dataset'
    checkpoint_dir = 'model'
    checkpoint_file = 'checkpoint.pth.tar'
    summary_dir = 'board'

    pretraining_step_size = 220

    # ---- MI355X build extras (absent in the reference) ----
    seed = None               # None -> random.randint(1, 10000) like the reference
    grad_bucket_mb = 64       # RCCL all-reduce bucket size
    log_file = 'train_epoch.log'
    num_workers = 1           # loader workers like the reference (agent/barGen2.py:41); spawned, never forked, persistent; 0 = in-process
    packed_data_file = None   # e.g. 'data/bars_packed.npz' (data.bar_dataset.pack_dataset): bit-packed rolls, expanded on the GPU
    compute_dtype = 'f32'     # 'bf16': bf16 matrix operands / fp32 accumulate in the conv kernels (tensors, master weights
                              # and Adam stay fp32) -- BASELINE.json configs 3-4
