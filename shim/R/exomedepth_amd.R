## exomedepth_amd.R -- the R functions on top of the cohort-level .Call entries of shim/edcore_shim.c.
##
## Drop this file into the R/ directory of the ExomeDepth package next to the reference's own sources (which stay as they
## are: R/class_definition.R:184-189 and R/tools.R:97 keep calling "get_loglike_matrix" and "C_hmm", now resolved by the
## shim).  R is not installed in the image this repository is built in, so nothing here has been run by R itself; what IS
## checked on every test run (tests/test_shim.py::test_r_wrappers_match_the_registered_entries) is that every
## .Call("name", ...) below names an entry the compiled shim registers and passes exactly as many arguments as that
## entry's registered arity (R's own check at call time, reference src/ExomeDepth_init.c:14-24).

## Every sample of a cohort at once: reference sets (vignette/vignette.Rnw:390-402), model fit (R/class_definition.R:82-191)
## and CNV calls (R/class_definition.R:311-419).
##   counts      integer matrix, exons x samples (columns named by sample), rows in any order
##   emit.mode   0: GSL's arithmetic operation for operation; 2 (default): table-driven emissions, sample-major -- R's
##               column-major matrix is uploaded as it lies (include/exomedepth_amd.h: ed_batch_set_emit_mode)
##   devices     NULL (default): every GPU of the node -- the samples are independent, so the slabs are dealt to the
##               devices from one queue, one host thread per device inside the .Call (ed_multi_*); or an integer
##               vector of device ordinals (0-based).  The result does not depend on it
##   slab        samples per slab of the pipeline (the unit the devices take from the queue and the fit's histograms are
##               laid out for): results are reproducible for a given slab whatever the node
CallCNVs.cohort <- function(counts, chromosome, start, end, name, transition.probability = 1e-4, expected.CNV.length = 50000,
                            n.bins.reduced = 10000, fit.mode = 0L, phi.bins = 1L, emit.mode = 2L, devices = NULL, slab = 256L) {
  if (length(start) != length(chromosome) || length(end) != length(chromosome) || length(name) != length(chromosome))
    stop('Chromosome, name, start and end vector must have the same lengths.\n')          # R/class_definition.R:319
  if (nrow(counts) != length(chromosome)) stop('The annotation vectors must have the same length as the rows of counts')
  ## exon order of CallCNVs (R/class_definition.R:322-327): levels from the names AS GIVEN -- "1".."22" first when present,
  ## every other name (X, Y, chr1, ...) in order of first appearance -- then the mid-point within a chromosome
  chr <- as.character(chromosome)
  chr.names.used <- unique(chr)
  chr.levels <- c(as.character(seq(1, 22)), chr.names.used[!chr.names.used %in% as.character(seq(1, 22))])
  chr.levels <- chr.levels[chr.levels %in% chr.names.used]
  o <- order(factor(chr, levels = chr.levels), 0.5 * (start + end))
  counts <- counts[o, , drop = FALSE]; chr <- chr[o]; start <- start[o]; end <- end[o]; name <- name[o]
  storage.mode(counts) <- "integer"
  chrom.off <- as.integer(c(0L, cumsum(table(factor(chr, levels = chr.levels)))))
  ## select.reference.set for every sample against all the others + the aggregate references (vignette.Rnw:390-402)
  rs <- .Call("ed_cohort_reference_sets", counts, as.double((end - start) / 1000), as.integer(n.bins.reduced), 32L,
              PACKAGE = "ExomeDepth")
  ## new('ExomeDepth') + CallCNVs() for every sample
  r <- .Call("ed_call_cnvs_batch", counts, rs$reference, chrom.off, as.integer(start), as.integer(end),
             as.double(transition.probability), as.double(expected.CNV.length), NULL, NULL, 1.0, as.integer(slab), 0L,
             as.integer(fit.mode), as.integer(phi.bins), as.integer(emit.mode),
             if (is.null(devices)) NULL else as.integer(devices), PACKAGE = "ExomeDepth")
  calls <- data.frame(sample = colnames(counts)[r$sample], start.p = r$start.p, end.p = r$end.p,
                      type = c("deletion", "duplication")[r$type], nexons = r$nexons,       # R/class_definition.R:385
                      start = start[r$start.p], end = end[r$end.p], chromosome = chr[r$start.p],  # :379-381
                      BF = r$BF, reads.expected = r$reads.expected, reads.observed = r$reads.observed,
                      reads.ratio = r$reads.ratio, stringsAsFactors = FALSE)
  calls$id <- paste('chr', calls$chromosome, ':', calls$start, '-', calls$end, sep = '')        # :383
  calls$id <- gsub(pattern = "chrchr", replacement = "chr", calls$id)                           # :384
  list(calls = calls, phi = r$phi, expected = r$expected, phi.bins = r$phi.bins, complete.bins = r$complete.bins,
       reference.choice = rs$choice, n.chosen = rs$n.chosen, n.unconverged = r$n.unconverged)
}

## What stands where new('ExomeDepth') calls aod::betabin (R/class_definition.R:118-119, :168), for every column at once.
## fit.mode 0: maximum likelihood; 1: aod's procedure (Nelder-Mead from the glm start) -- a point inside optim()'s
## tolerance region, not pinned against aod itself.  Returns list(phi, expected, converged).
fit.betabin.cohort <- function(test, reference, fit.mode = 0L) {
  storage.mode(test) <- "integer"; storage.mode(reference) <- "integer"
  .Call("ed_fit_betabin_batch", test, reference, as.integer(fit.mode), PACKAGE = "ExomeDepth")
}

## select.reference.set (R/optimize_reference_set.R:53-148) for one test sample on the GPU.
select.reference.set.gpu <- function(test.counts, reference.counts, bin.length = NULL, n.bins.reduced = 0L) {
  storage.mode(reference.counts) <- "integer"
  r <- .Call("ed_select_reference_set", as.integer(test.counts), reference.counts,
             if (is.null(bin.length)) NULL else as.double(bin.length), as.integer(n.bins.reduced), PACKAGE = "ExomeDepth")
  r
}

## ... for every sample of a cohort against all the others, with the aggregate reference of every sample
## (the loop of vignette/vignette.Rnw:390-402).  Returns list(n.chosen, choice, reference, correlations, n.bins).
select.reference.set.cohort <- function(counts, bin.length = NULL, n.bins.reduced = 10000L, max.refs = 32L) {
  storage.mode(counts) <- "integer"
  .Call("ed_cohort_reference_sets", counts, if (is.null(bin.length)) NULL else as.double(bin.length),
        as.integer(n.bins.reduced), as.integer(max.refs), PACKAGE = "ExomeDepth")
}
