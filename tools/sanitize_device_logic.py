#!/usr/bin/env python3
"""tools/sanitize_device_logic.py -- the per-fragment / per-candidate device logic (arriba_amd/csrc/device/*_core.hpp), stepped on the host by
tests/emu, under AddressSanitizer + UBSan against the golden dumps (test tooling).  Run through tools/sanitize_device_logic.sh, which builds
the instrumented harness and preloads the sanitizer runtimes into Python."""
import sys, os, ctypes
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import parity, conftest, datasets, tempfile
from arriba_amd import _capi
api=_capi.bind_device_api(ctypes.CDLL(os.environ['ARRIBA_EMU_LIBRARY']),'emu_')
tmp=tempfile.mkdtemp(prefix='asan_')
for name in ['toy3k','stacked4k','itd6k']:
    prefix=datasets.generate(datasets.DATASETS[name], tmp, name)
    golden=conftest.golden_dir(name)
    s,p=parity.run_read_level(parity.open_session,prefix,api=api)
    if name!='itd6k':
        parity.check_read_filters(s,p,golden); parity.check_annotation(s,p,golden)
    p.find_fusions(); print(name,'candidates',parity.check_candidates(s,p,golden))
    if name=='toy3k':
        print('evalues',parity.check_evalues(s,p,golden)); print('mismappers',parity.check_mismappers(s,p,golden))
        s,p=parity.run_read_level(parity.open_session,prefix,api=api)
        print('multimappers',parity.check_multimappers(s,p,golden))
        s,p=parity.run_read_level(parity.open_session,prefix,api=api)
        print('chain',parity.check_chain_to_relative_support(s,p,golden,multimappers=True))
        s,p=parity.run_read_level(parity.open_session,prefix,api=api)
        print('event predicates',parity.check_event_predicates(s,p,golden))
        s,p=parity.run_read_level(parity.open_session,prefix,api=api)
        print('event chain',parity.check_event_chain(s,p,golden))
    if name=='itd6k':
        p.merge_adjacent_fusions(); print('merged lists',parity.check_read_lists(s,p,golden,'merge_adjacent_fusions')); print('recover_itd',parity.check_recover_itd(s,p,golden))
        s,p=parity.run_read_level(parity.open_session,prefix,api=api)
        print('chain to no_coverage',parity.check_chain_to_no_coverage(s,p,golden))
for name, check in (('homologs8k_open', lambda s,p,g: parity.check_homologs(s,p,g,state_from='recover_many_spliced')), ('homologs8k', parity.check_chain_to_isoforms)):
    prefix=datasets.generate(datasets.DATASETS[name], tmp, name)
    s,p=parity.run_read_level(parity.open_session,prefix,api=api)
    print(name, check(s,p,conftest.golden_dir(name)))
prefix=datasets.generate(datasets.DATASETS['toy3k'], tmp, 'toy3k_confidence')
s,p=parity.run_read_level(parity.open_session,prefix,api=api)
print('confidence', parity.check_confidence(s,p,conftest.golden_dir('toy3k')))
prefix=datasets.generate(datasets.DATASETS['rules8k'], tmp, 'rules8k')
s,p=parity.run_read_level(parity.open_session,prefix,api=api)
print('range rules', parity.check_range_rules(s,p,conftest.golden_dir('rules8k'),prefix))
for name in ('toy3k', 'rules8k'):  # the whole chain and the output writer (host library built with the sanitizers, too)
    prefix=datasets.generate(datasets.DATASETS[name], tmp, name+'_out')
    s,p=parity.run_read_level(parity.open_session,prefix,api=api)
    parity.check_chain_to_isoforms(s,p,conftest.golden_dir(name),rules_prefix=prefix if name=='rules8k' else None)
    os.makedirs(os.path.join(tmp,name+'_files'))
    print('output files', name, parity.check_output_files(s,p,conftest.golden_dir(name),os.path.join(tmp,name+'_files'),rules_prefix=prefix if name=='rules8k' else None))
prefix=datasets.generate(datasets.DATASETS['wgs8k'], tmp, 'wgs8k')
os.makedirs(os.path.join(tmp,'wgs8k_files'))
print('workflow with -d', parity.check_workflow(prefix,conftest.golden_dir('wgs8k'),os.path.join(tmp,'wgs8k_files'),api=api,rules=True,structural_variants=True)[-1])
# the device ingest, stepped (emu_ingest.inc over ingest_core.hpp: records read in 8-byte words, the ITD probe from two words, coverage_t as ranges, the hit index kept per record):
# the batch of the host ingest, on datasets with tandem duplications, multi-mappers, duplicates, long names; then the whole workflow from the bytes of the file
import subprocess
import test_host_and_device_logic as cpu_tier
from arriba_amd.pipeline import DevicePipeline, HostSession
for name in ('toy3k', 'itd6k', 'shuffled_dups_40k', 'stranded_multimappers_20k'):
    prefix=datasets.generate({'args': cpu_tier.DEVICE_INGEST_DATASETS[name]}, tmp, name+'_ingest')
    host=HostSession(prefix+'.fa', prefix+'.gtf'); host.read_chimeric_alignments(prefix+'.bam')
    expected=cpu_tier._batch_columns(host); expected['coverage']=int(host._lib.ahost_coverage_checksum(host._session))
    session=HostSession(prefix+'.fa', prefix+'.gtf')
    columns=cpu_tier._device_batch_columns(session, DevicePipeline(session, api=api, bam=prefix+'.bam', piece_bytes=1<<20))
    print('device ingest', name, [key for key in expected if expected[key]!=columns[key]] or 'equal')
long_names=os.path.join(tmp,'long')
subprocess.run([datasets.GEN_SYNTH,'--out',long_names,'--seed','11','--fragments','3000','--contigs','4','--contig-len','300000','--junctions','60','--name-length','45'],check=True,stdout=subprocess.DEVNULL,stderr=subprocess.DEVNULL)
host=HostSession(long_names+'.fa', long_names+'.gtf'); host.read_chimeric_alignments(long_names+'.bam')
session=HostSession(long_names+'.fa', long_names+'.gtf')
columns=cpu_tier._device_batch_columns(session, DevicePipeline(session, api=api, bam=long_names+'.bam', piece_bytes=1<<20))
print('device ingest, 45-character names', [key for key, value in cpu_tier._batch_columns(host).items() if value!=columns[key]] or 'equal')
prefix=datasets.generate(datasets.DATASETS['toy3k'], tmp, 'toy3k_from_bytes')
os.makedirs(os.path.join(tmp,'from_bytes'))
print('workflow from the bytes of the file', parity.check_workflow(prefix,conftest.golden_dir('toy3k'),os.path.join(tmp,'from_bytes'),api=api,device_ingest=True)[-1])
print('done')
