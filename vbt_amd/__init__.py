"""vbt_amd: the MI355X-native hot path of simonkosina/vbt (reference track.py:159-247, plot.py:33-47).

Importing the package touches neither the environment nor the GPU.  The pipeline keeps four HIP streams busy (three forwards in
flight + the copy stream of the host-fed mode) and the GPU serves four hardware queues side by side, one per compute pipe - HIP's own
default for GPU_MAX_HW_QUEUES.  A process that sets the variable to something else gets a pipeline whose placement check
(vbt_pipeline_create, csrc/pipeline.hip) reports the streams it could not separate, instead of this package overriding the setting."""
