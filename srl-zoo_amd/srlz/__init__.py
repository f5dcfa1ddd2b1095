"""srlz — Python host binding of the MI355X kernels (libsrlz_hip.so) behind srl-zoo's training hot path."""
