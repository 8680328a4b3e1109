% compile_hip.m -- builds the three MEX gateways against libfftconv.so; the counterpart of the
% reference's compile.m / cuda_compile.m (compile.m:9-11 lists the same three entry points).
% Run from the repository root on a host with MATLAB, ROCm and an MI355X:
%   >> run('cuda-fft-convolution_amd/mex/compile_hip.m')
% The HIP kernels are compiled by `make -C cuda-fft-convolution_amd/csrc` (hipcc, gfx950); the
% gateways themselves are plain C++ and only need MATLAB's own compiler driver.
repo = pwd;
pkg = fullfile(repo, 'cuda-fft-convolution_amd');
if ~exist(fullfile(pkg, 'libfftconv.so'), 'file')
  assert(system(['make -C ' fullfile(pkg, 'csrc')]) == 0, 'building libfftconv.so failed');
end
if ~exist(fullfile(repo, 'bin'), 'dir'), mkdir(fullfile(repo, 'bin')); end
names = {'cudaFFTData', 'cudaConvFFTData', 'cudaConvolutionFFT', 'cudaConvFFTDataStreams'};   % the last needs the GPU header
% gpuArray kernels (src/cudaConvolutionFFT.cu:224-238): the gateways compile that branch when
% gpu/mxGPUArray.h is found; it lives where cuda_compile.m:48-52 takes it from and needs libmwgpu (:58)
gpuinc = fullfile(matlabroot, 'toolbox', 'distcomp', 'gpu', 'extern', 'include');
gpuargs = {};
if exist(fullfile(gpuinc, 'gpu', 'mxGPUArray.h'), 'file'), gpuargs = {['-I' gpuinc], '-lmwgpu'}; end
for i = 1:numel(names)
  mex('-largeArrayDims', ['-I' fullfile(repo, 'include')], gpuargs{:}, ...
      fullfile(pkg, 'mex', [names{i} '_mex.cpp']), ...
      ['-L' pkg], '-lfftconv', ['LDFLAGS=$LDFLAGS -Wl,-rpath,' pkg], ...
      '-output', fullfile(repo, 'bin', names{i}));
end
addpath(fullfile(repo, 'bin'));
